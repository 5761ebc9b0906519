"""cwf -- host-side core of the MI355X-native ClsWiseFormer hot path: ctypes binding of libcwf_hip.so
(_lib), tensor-level kernel wrappers (kernels), autograd glue (functional), weight/gradient layout maps
(packing), fused optimizer (optim), gradient all-reduce (parallel) and the training-step harness (trainer)."""
