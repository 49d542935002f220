"""`python -m dfmdock_amd dock | sweep | selfcheck ...` - see dfmdock_amd/cli.py."""
import sys

from .cli import main

sys.exit(main())
