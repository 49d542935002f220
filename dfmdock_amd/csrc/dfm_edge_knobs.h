// dfm_edge_knobs.h - every compile-time switch of kernels_edge.hip in one place: the shipped values and what each one measured.
// A variant is built next to the product library with tools/build_variant.sh NAME "-DDFM_EDGE_...=v" and compared on one box with
// tools/ab_lib.sh (DFM_LIB selects the library); none of them is read at run time.  Experiments that were concluded and removed
// from the source keep their records under profiles/ (dense row packing ceiling: r03_exp_dense_rows.txt; unpacked fp32 arithmetic:
// r02_exp_edge_trims.txt; 16-row tiles: r03_exp_tile16.txt; fused last layer: r03_exp_last_fused.txt).
#pragma once

// ---- scheduling of the message kernel's chunk pipeline -------------------------------------------------------------------------
#ifndef DFM_EDGE_BD
#define DFM_EDGE_BD 2
#endif
#ifndef DFM_EDGE_SB
#define DFM_EDGE_SB 1      // a scheduling barrier after every DFM_EDGE_SB-th MFMA slot of a chunk
#endif
// Slots (MFMA index inside a chunk) after which the producer requests the next-but-one chunk's operands.  The vector-memory
// counter completes in order, so a wait for the YOUNGEST load a slot needs also waits for everything issued before it:
// the per-chunk constants (A_i, w_r: L1 hits, used from slot 0 of the next chunk) go out BEFORE the second pass's gathers (L2
// hits, used from slot 8), and both gathers as early as their registers are dead (pass registers die after the pass's slice 0).
#ifndef DFM_EDGE_G0
#define DFM_EDGE_G0 7
#endif
#ifndef DFM_EDGE_G1
#define DFM_EDGE_G1 15
#endif
#ifndef DFM_EDGE_GC
#define DFM_EDGE_GC 14
#endif
#ifndef DFM_EDGE_DEFER      // requests of the next tile's chunk 1 issued after the epilogue instead of under chunk 7: 0 none, 1 A_i / w_r, 2 + second pass
#define DFM_EDGE_DEFER 2
#endif
#ifndef DFM_EDGE_WAVES      // waves per workgroup: 8 = two per SIMD (256 registers each), 4 = one per SIMD (512)
#define DFM_EDGE_WAVES 8
#endif

// r06 experiments on the producer's dependent chains (profiles/r06_exp_edge_sched.txt):
#ifndef DFM_EDGE_ILV        // 1: the two 16-row passes of a chunk interleaved slot by slot (both passes' operands needed from slot 0 / 1)
#define DFM_EDGE_ILV 0
#endif
#ifndef DFM_EDGE_SKEW       // 1: a pair's two v_exp go out at the end of its pre-activation slice, one MFMA slot ahead of the rest of its SiLU
#define DFM_EDGE_SKEW 0
#endif
#ifndef DFM_EDGE_PRIO       // n > 0: s_setprio n for waves 4..7 (the second-dispatched wave of every SIMD)
#define DFM_EDGE_PRIO 0
#endif
#ifndef DFM_EDGE_ROT        // 1: a slot is {producer slice, MFMA} instead of {MFMA, slice}: the chunk's first MFMA no longer follows its fragment reads directly
#define DFM_EDGE_ROT 0
#endif
#ifndef DFM_EDGE_NOFENCE    // 1: no s_waitcnt lgkmcnt(0) between the producer's ds_write and the fragment reads of the same wave (LDS executes a wave's instructions in order)
#define DFM_EDGE_NOFENCE 0
#endif
#ifndef DFM_EDGE_EARLYA     // 1: k-step 0's A fragment and the first weight fragment of chunk c + 1 are requested at the end of chunk c (chunks 1..7)
#define DFM_EDGE_EARLYA 0
#endif
#ifndef DFM_EDGE_MSTORE_NOFENCE   // 1: the transposing stores through the staging buffer rely on in-order LDS execution instead of s_waitcnt lgkmcnt(0) twice per n-tile
#define DFM_EDGE_MSTORE_NOFENCE 0
#endif
#ifndef DFM_EDGE_PAD0       // 1: the padded rows of a node's second tile (60 -> 64) gather node 0's Bm row (one hot line per trajectory) instead of the node's own
#define DFM_EDGE_PAD0 0
#endif
#ifndef DFM_EDGE_KO         // knock-out builds, WRONG RESULTS: bit 0 no MFMA, bit 1 no producer transcendentals, bit 2 no epilogue transcendentals, bit 3 one weight-fragment LDS read per chunk
#define DFM_EDGE_KO 0
#endif

// ---- memory-side choices ---------------------------------------------------------------------------------------------------------
#ifndef DFM_EDGE_NT         // 1: single-pass streams (edge data, agg / message stores) carry the non-temporal hint
#define DFM_EDGE_NT 1
#endif
#ifndef DFM_EDGE_A_NT       // 1: the A_i row carries that hint too; 0 (shipped): cached - the node's second tile and the other half of each
#define DFM_EDGE_A_NT 0     // 128-byte line re-read it: 2.187 vs 2.221 ms per launch, same box
#endif
#ifndef DFM_EDGE_MSTORE_LDS // last layer: gated messages transposed through the wave's staging buffer (1) or stored as 2-byte scatters (0)
#define DFM_EDGE_MSTORE_LDS 1
#endif
// DFM_TAB_MERGE (dfm_internal.h; shared with api.hip's table fold): 1 = two merged lookup tables per layer, 0 = three

// ---- diagnostic builds (WRONG RESULTS or extra output by design; never defined in the product build) --------------------------------
//   -DDFM_EDGE_STAMP       s_memtime stamps of the tile phases into EdgeKArgs::stamp (tools/edge_phases.py)
//   -DDFM_F32M_STAMP       the same for k_edge_f32m
//   -DDFM_EDGE_SAMEROW=n   every row gathers row 0 (loads issued, L1 hits): 1 everything, 2 Bm only, 3 tables only
//   -DDFM_EDGE_NOGATHER    the gathered operands are whatever the registers hold (no gather instructions)
