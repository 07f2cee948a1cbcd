"""decombinator_amd — MI355X-native `decombine` hot path of Decombinator.

csrc/        HIP kernels + the C ABI (libdcrx.so, include/dcrx.h)
_native.py   ctypes binding of the library
decombine.py host mirror of the reference's decombine stage (same entry points)
io.py        argument dictionary / CLI parser / .n12 writer
pipeline.py  sub-command dispatch
sharded.py   multi-GPU sharding and the RCCL gather of DCR tuples
synth.py     synthetic tag sets in the reference's file formats
"""
