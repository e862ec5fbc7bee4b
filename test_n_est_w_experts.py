#!/usr/bin/env python3
"""Drop-in for the reference script of the same name: same flags, same .xyz in, same
.normals / .experts / .experts_probs out.  All work happens in nesti_net_amd (HIP kernels)."""
import os
import sys

if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd.cli import main
    sys.exit(main())
