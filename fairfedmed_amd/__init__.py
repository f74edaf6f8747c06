"""MI355X-native FairLoRA local-training engine."""
import os as _os

# Three engine streams + RCCL's need more than ROCm's default 4 hardware queues to overlap; read by the
# HIP runtime when it initialises, so it has to be in the environment before the first GPU call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
