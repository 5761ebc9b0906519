"""cwf -- host-side core of the MI355X-native ClsWiseFormer hot path: ctypes binding of libcwf_hip.so
(_lib), tensor-level kernel wrappers (kernels), autograd glue (functional), weight/gradient layout maps
(packing), fused optimizer (optim), gradient all-reduce (parallel) and the training-step harness (trainer)."""

import os as _os

# The HIP runtime multiplexes ALL streams of a process over GPU_MAX_HW_QUEUES hardware queues (default 4), and a stream that waits for an
# event blocks its whole queue.  A data-parallel step uses the main stream, the weight-gradient side stream, the communication stream
# and RCCL's own streams: with four queues they share, and the cross-stream waits of the launch plan serialise -- 27.6 ms per step
# instead of 18.2 with nothing but a one-rank RCCL group's collectives enabled (tools/r3_comm.sh).  Eight queues restore it (18.3 ms).
# Must be in the environment before the process's first HIP call, hence here (importing this package precedes any launch).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
