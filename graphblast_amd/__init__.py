"""graphblast_amd -- MI355X (gfx950) backend for the GraphBLAST mxv/vxm hot path.

The package is a thin host layer over libgrb_hip.so (C ABI in include/grb_hip.h):
  api       Vector / Matrix / Descriptor and vxm, mxv, eWiseAdd, eWiseMult, reduce, assign,
            bfs, sssp, pr with the reference frontend's names and error behaviour
  graphgen  seeded synthetic inputs (RMAT, grid) and the loader pipeline feeding them
  dist      1-D vertex partitioning across the GPUs of a node (torch.distributed / RCCL)
Importing it loads the HIP library and raises if it is missing: there is no CPU path.
"""
from . import _lib

_lib.load()

from .api import *  # noqa: E402,F401
from . import api  # noqa: E402,F401
from . import graphgen  # noqa: E402,F401
