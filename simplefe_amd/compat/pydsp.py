"""`from pydsp import *` for the reference's driver scripts (libdsp/test/test_decimate.py:8,
test_resample.py): put this directory on sys.path in place of the SWIG build directory
(`sys.path.append('../build/test')` there) and the names the SWIG module exports
(libdsp/test/pydsp.i:21-22: resample, decimate; blkconv added) resolve to the GPU classes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplefe_amd.api import blkconv, decimate, resample  # noqa: E402,F401

__all__ = ["resample", "decimate", "blkconv"]
