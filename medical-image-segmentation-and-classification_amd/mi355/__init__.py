"""MI355X launch-plan engine."""
import os

# The engine overlaps weight-gradient kernels (side stream) and gradient all-reduces (comm stream) with the main stream; the
# HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and with RCCL's streams in the process two
# of ours end up sharing one, which serialises them (+0.7 ms on a 19 ms step).  Takes effect only if the HIP runtime has not
# initialised yet, i.e. when this package is imported before the first CUDA call; an explicit setting wins.
import sys

HW_QUEUES_EXPLICIT = "GPU_MAX_HW_QUEUES" in os.environ
# (best effort: PyTorch's own lazy initialisation is what can be asked; a bare hip call made elsewhere earlier cannot be seen)
HW_QUEUES_IN_TIME = HW_QUEUES_EXPLICIT or not ("torch" in sys.modules and sys.modules["torch"].cuda.is_initialized())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
