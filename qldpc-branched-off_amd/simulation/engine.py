"""``run_simulation`` with the reference's signature and result keys (src/simulation/engine.py:193-488), batched on GPUs.

The reference spawns a process pool and runs one trial per task (engine.py:433-457); here all trials of a batch run
concurrently on the device (qldpc_circuit_plan_*), the tally stays on the device, and with several ranks
(torch.distributed) each rank takes a contiguous trial range followed by ONE all-reduce of the tally.

Differences that are deliberate and documented:
  * randomness comes from Philox streams keyed by (base_seed, global trial index), not from legacy ``np.random``; results
    are reproducible and independent of the number of GPUs, and statistically equivalent to the reference;
  * ``osd_order`` must be 0 (OSD-w sweep is a 'next' row); alpha estimation (alpha.py / scopt.py) is out of scope, so
    ``alpha_mode='alvarado'`` needs ``alvarado_alpha`` and ``'alvarado-autoregressive'`` is not available;
  * ``target_logical_errors`` stops at batch granularity (the reference stops at trial granularity, engine.py:462-464).
"""
import numpy as np

from .. import _lib, parallel
from ..codes.bb_code import BBCodeCircuit
from ..noise.compiled import CompiledCircuit
from ..noise.builder import build_decoding_matrices


def prior_llrs(channel_probs):
    """engine.py:210-212: clip(nan_to_num(log((1 - p) / p)), -50, 50) (p_j > 1 -> NaN -> 0)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.clip(np.nan_to_num(np.log((1 - channel_probs) / channel_probs)), -50, 50)


def run_simulation(Hx, Hz, Lx, Lz, error_rate, num_trials=1000, num_cycles=12, maxIter=50, osd_order=0, use_dynamic_alpha=True,
                   alpha_mode=None, alvarado_alpha=None, alpha_estimation_trials=5000, alpha_estimation_bins=50, precomputed_matrices=None,
                   num_workers=None, base_seed=None, use_jit=True, target_logical_errors=None, max_trials=None, scopt=False,
                   estimation_plot_dir=None, batch=4096, device=0, **bb_params):
    if osd_order != 0:
        raise NotImplementedError("OSD-w with w > 0 is a 'next' row; use osd_order=0")
    if scopt:
        raise NotImplementedError("SCOPT beta estimation is out of scope (unused by the reference decoder, engine.py:389)")
    if base_seed is None:
        base_seed = int(np.random.randint(0, 2 ** 31))
    if alpha_mode is None:
        alpha_mode = "dynamical" if use_dynamic_alpha else "alvarado"
    if alpha_mode == "alvarado":
        if alvarado_alpha is None:
            raise NotImplementedError("alpha estimation (src/decoding/alpha.py) is out of scope: pass alvarado_alpha")
        if isinstance(alvarado_alpha, (list, tuple, np.ndarray)) and len(alvarado_alpha) == 2:
            alpha_z, alpha_x = float(alvarado_alpha[0]), float(alvarado_alpha[1])
        else:
            alpha_z = alpha_x = float(alvarado_alpha)
    elif alpha_mode == "dynamical":
        alpha_z = alpha_x = 1.0
    elif alpha_mode == "alvarado-autoregressive":
        raise NotImplementedError("autoregressive alpha estimation is out of scope")
    else:
        raise ValueError(f"Unsupported alpha_mode: {alpha_mode}")

    cb = BBCodeCircuit(Hx, Hz, num_cycles=num_cycles, **bb_params)
    m = precomputed_matrices or build_decoding_matrices(cb, Lx, Lz, error_rate, verbose=False)          # engine.py:207-208
    compiled = CompiledCircuit(base_circuit=cb.get_full_circuit(), noiseless_suffix=cb.cycle * 2, lin_order=cb.lin_order,
                               data_qubits=cb.data_qubits, Xchecks=cb.Xchecks, Zchecks=cb.Zchecks)
    llrs_z, llrs_x = prior_llrs(np.asarray(m["channel_probsZ"], dtype=np.float64)), prior_llrs(np.asarray(m["channel_probsX"], dtype=np.float64))
    k = np.asarray(Lx).shape[0]
    graphs, masks = [], []
    for s in ("Z", "X"):
        Hdec = m[f"Hdec{s}"]
        ip, ix, shape = _lib.canonical_csr(Hdec)
        graphs.append(_lib.Graph(ip, ix, shape[1], device=device))
        if f"H{s}_logical" in m:                     # compact form: the k logical rows only (dense or (indptr, indices))
            masks.append(_lib.logical_column_masks(m[f"H{s}_logical"], shape[1]))
        else:
            flr = int(m[f"first_logical_row{s}"])
            masks.append(_lib.logical_column_masks(np.asarray(m[f"H{s}_full"])[flr:flr + k], shape[1]))    # engine.py:412-413
    plan = _lib.CircuitPlan(compiled, Lx, Lz, graphs[0], graphs[1], llrs_z, llrs_x, masks[0], masks[1], error_rate, max_iter=maxIter,
                            alpha_z=alpha_z, alpha_x=alpha_x, alpha_mode=alpha_mode, use_osd=True, batch=batch)

    if max_trials is None:
        max_trials = num_trials if num_trials is not None else 1000000
    stop_on_errors = target_logical_errors is not None and target_logical_errors > 0
    rank, world = 0, 1
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    T = _lib.TALLY
    total = np.zeros(_lib.TALLY_SLOTS, np.int64)
    done = 0
    round_size = batch * world if stop_on_errors else max_trials
    while done < max_trials:
        this = min(round_size, max_trials - done)
        begin, count = parallel.shard_range(this, rank, world)
        if count:
            plan.run(base_seed, done + begin, count)
        total += parallel.allreduce_tally(plan.read(clear=True))                    # engine.py:450-457
        done += this
        if stop_on_errors and total[T["total_err"]] >= target_logical_errors:        # engine.py:462-464
            break
    plan.close()
    return parallel.tally_to_result(total) | {"tally": total}
