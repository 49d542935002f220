"""dfmdock_amd: MI355X-native sampling engine for DFMDock-style rigid docking.

Only the sampling hot path lives here: HIP kernels + C ABI
under ``csrc/`` and the host-side mirror of the reference's Python interface.
"""
__version__ = "0.1.0"
