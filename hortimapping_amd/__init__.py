"""MI355X-native HortiMapping hot path.  Importing the package fixes the one process-wide setting the
multi-GPU path needs BEFORE the HIP / HSA runtime starts (it is read once, at runtime start-up): dmabuf IPC for RCCL
between processes.  A value the caller exported wins; `hortimapping_amd.distributed` warns when it was too late."""
import os as _os

_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
