"""MI355X-native accelerator for CoEvoNet's population-evaluation hot path."""
import os as _os

# More hardware queues for the cohort streams (HIP's default of 4 lets a later engine's cohort stream share the caller's
# queue, which serialises its launches); only effective when set before the HIP runtime starts - bench.py sets it itself.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
