"""Circuit-level noise: same call surface as the reference's ``src/noise`` package (JIT path), computed on the GPU."""
from .simulation import run_trial_fast  # noqa: F401
from .compiled import CompiledCircuit  # noqa: F401
from .builder import build_decoding_matrices  # noqa: F401
