"""``run_simulation`` with the reference's signature and result keys (src/simulation/engine.py:193-488), batched on GPUs.

The reference spawns a process pool and runs one trial per task (engine.py:433-457); here all trials of a batch run
concurrently on the device (qldpc_circuit_plan_*), the tally stays on the device, and with several ranks
(torch.distributed) each rank takes a contiguous trial range followed by ONE all-reduce of the tally.

Differences that are deliberate and documented:
  * randomness comes from Philox streams keyed by (base_seed, global trial index), not from legacy ``np.random``; results
    are reproducible and independent of the number of GPUs, and statistically equivalent to the reference;
  * the alpha / beta estimators (``alpha_mode='alvarado*'``, ``scopt=True``) draw from ``default_rng(base_seed)`` instead of an
    unseeded Generator, so all ranks agree on the factors; their trial loops run on the GPU (qldpc_msgstats_*);
  * ``osd_order > 0``: the reference returns the OSD-0 solution whenever it reproduces the syndrome (osd.py:27-29), which it always
    does for syndromes the circuit itself generates -- those trials stay on the fused device pipeline.  A batch in which some trial
    ends unsatisfied (possible with foreign ``precomputed_matrices``) is decoded again through the batched OSD-w sweep
    (qldpc_osdw_batch, osd.py:31-75) for the shots BP failed on, so the result is the reference's for every input;
  * ``num_workers`` keeps the reference's meaning -- how many workers ONE call spreads its trials over (engine.py:204,433-435) -- with a worker
    being a GPU instead of a pool process: the call creates one plan per worker (its own graphs, buffers and stream on its device), hands every
    worker a contiguous trial range per round from a host thread each (the C ABI releases the GIL), and sums the tallies on the host.
    ``num_workers=None`` = every visible GPU (1 on a one-GPU box), a larger request is capped at the visible count; ``devices=[...]`` (extension)
    names the GPU of each worker explicitly and may repeat one ([0, 0] = two plans sharing a card).  Results do not depend on it;
  * one process per GPU also works: under a ``torch.distributed`` launch a rank drives ``device`` = LOCAL_RANK alone (``num_workers`` then counts
    the plans of that rank, default 1) and one all-reduce of the tally follows; with the RCCL backend it runs on device tensors;
  * ``target_logical_errors`` stops at the exact trial the reference would (in-order prefix cut over per-trial verdicts).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .. import _lib, parallel
from ..decoding.alpha import estimate_alpha_alvarado, estimate_alpha_alvarado_autoregressive
from ..decoding.scopt import estimate_scopt_beta
from ..codes.bb_code import BBCodeCircuit
from ..noise.compiled import CompiledCircuit
from ..noise.builder import build_decoding_matrices


def prior_llrs(channel_probs):
    """engine.py:210-212: clip(nan_to_num(log((1 - p) / p)), -50, 50) (p_j > 1 -> NaN -> 0)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.clip(np.nan_to_num(np.log((1 - channel_probs) / channel_probs)), -50, 50)


def _estimation_trials(requested, n_cols, error_rate):
    """engine.py:230-246: enough trials for ~2000 flipped bits, clamped to [500, 50000], unless the caller changed the default."""
    dynamic = max(500, min(50000, int(2000 / (n_cols * error_rate))))
    return requested if requested != 5000 else dynamic


def run_simulation(Hx, Hz, Lx, Lz, error_rate, num_trials=1000, num_cycles=12, maxIter=50, osd_order=0, use_dynamic_alpha=True,
                   alpha_mode=None, alvarado_alpha=None, alpha_estimation_trials=5000, alpha_estimation_bins=50, precomputed_matrices=None,
                   num_workers=None, base_seed=None, use_jit=True, target_logical_errors=None, max_trials=None, scopt=False,
                   estimation_plot_dir=None, batch=16384, device=None, flags=0, devices=None, **bb_params):
    if osd_order < 0:
        raise ValueError("osd_order must be >= 0")
    if num_workers is not None and int(num_workers) < 1:
        raise ValueError("num_workers must be >= 1")
    rank, world = 0, 1
    try:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            rank, world = _dist.get_rank(), _dist.get_world_size()
    except ImportError:
        _dist = None
    # which GPU each worker of THIS process drives (engine.py:204: num_workers)
    if devices is not None:
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one GPU")
    elif world > 1 or device is not None or "LOCAL_RANK" in os.environ:
        devices = [parallel.local_device(device)] * int(num_workers or 1)       # a rank of a multi-process launch: its own GPU
    else:
        visible = max(1, _lib.device_count())
        devices = list(range(min(int(num_workers), visible) if num_workers else visible))
    device = devices[0]
    if world > 1 and _dist.get_backend() == "nccl":
        import torch
        torch.cuda.set_device(device)              # RCCL collectives run on the rank's own GPU
    if base_seed is None:
        base_seed = int(np.random.randint(0, 2 ** 31))
    if alpha_mode is None:
        alpha_mode = "dynamical" if use_dynamic_alpha else "alvarado"
    if alpha_mode not in ("dynamical", "alvarado", "alvarado-autoregressive"):
        raise ValueError(f"Unsupported alpha_mode: {alpha_mode}")
    if alpha_mode == "alvarado-autoregressive" and alvarado_alpha is not None:
        raise ValueError("alvarado_alpha must be None for alvarado-autoregressive")                       # engine.py:294-295
    if estimation_plot_dir is not None:
        os.makedirs(estimation_plot_dir, exist_ok=True)

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

    # Normalisation factors (engine.py:228-344).  The estimators draw their error patterns from a Generator seeded with
    # base_seed (the reference leaves it unseeded), so every rank of a multi-GPU run derives the same factors.
    extra = {}
    est_rng = np.random.default_rng(base_seed)
    rate_tag = f"{error_rate:.6g}".replace(".", "p")
    if alpha_mode == "alvarado":
        if alvarado_alpha is None:
            fits = [estimate_alpha_alvarado(g, error_rate, trials=_estimation_trials(alpha_estimation_trials, g.n, error_rate),
                                            bins=alpha_estimation_bins, rng=est_rng, plot_dir=estimation_plot_dir,
                                            plot_prefix=f"alvarado_{rate_tag}_{tag}", llrs=llr)
                    for g, llr, tag in ((graphs[0], llrs_z, "z"), (graphs[1], llrs_x, "x"))]
            (alpha_z, r2_z), (alpha_x, r2_x) = fits
        elif isinstance(alvarado_alpha, (list, tuple, np.ndarray)) and len(alvarado_alpha) == 2:
            alpha_z, alpha_x, r2_z, r2_x = float(alvarado_alpha[0]), float(alvarado_alpha[1]), None, None
        else:
            alpha_z = alpha_x = float(alvarado_alpha)
            r2_z = r2_x = None
        extra.update(alpha_r2_z=r2_z, alpha_r2_x=r2_x)                                                      # engine.py:479-481
    elif alpha_mode == "alvarado-autoregressive":
        fits = [estimate_alpha_alvarado_autoregressive(g, error_rate, maxIter=maxIter, trials=_estimation_trials(alpha_estimation_trials, g.n, error_rate),
                                                       bins=alpha_estimation_bins, rng=est_rng, plot_dir=estimation_plot_dir,
                                                       plot_prefix=f"autoregressive_{rate_tag}_{tag}", llrs=llr)
                for g, llr, tag in ((graphs[0], llrs_z, "z"), (graphs[1], llrs_x, "x"))]
        (alpha_z, r2s_z), (alpha_x, r2s_x) = fits
        extra.update(alpha_values_z=alpha_z, alpha_values_x=alpha_x, alpha_r2_values_z=r2s_z, alpha_r2_values_x=r2s_x)   # engine.py:474-478
    else:
        alpha_z = alpha_x = 1.0
    if scopt:                                                                                                # engine.py:346-387 (beta is reported, not used)
        betas = [estimate_scopt_beta(g, error_rate, trials=_estimation_trials(5000, g.n, error_rate), bins=alpha_estimation_bins, alpha=a,
                                     alpha_mode=alpha_mode, maxIter=maxIter, rng=est_rng, plot_dir=estimation_plot_dir,
                                     plot_prefix=f"scopt_{rate_tag}_{tag}", llrs=llr)
                 for g, llr, a, tag in ((graphs[0], llrs_z, alpha_z, "z"), (graphs[1], llrs_x, alpha_x, "x"))]
        extra.update(beta_z=betas[0][0], beta_x=betas[1][0], beta_r2_z=betas[0][1], beta_r2_x=betas[1][1])   # engine.py:482-486

    T = _lib.TALLY
    csr = [(g.indptr, g.indices, g.n) for g in graphs]

    class Worker:
        """One worker = the reference's pool process (engine.py:433-435) as a device plan: graphs, masks, buffers and a stream of its own on one GPU."""

        def __init__(self, dev, own_graphs=None):
            self.device = dev
            self.graphs = own_graphs or [_lib.Graph(ip, ix, n, device=dev) for ip, ix, n in csr]
            self.stream = _lib.Stream(dev)
            self.plan = _lib.CircuitPlan(compiled, Lx, Lz, self.graphs[0], self.graphs[1], llrs_z, llrs_x, masks[0], masks[1], error_rate, max_iter=maxIter,
                                         alpha_z=alpha_z, alpha_x=alpha_x, alpha_mode=alpha_mode, use_osd=True, batch=batch, flags=flags)    # flags: QLDPC_FLAG_* kernel variants (extension)

        def osdw_batch(self, begin, count):
            """One trial range through sample -> decode -> OSD-w (order = osd_order) on the shots BP failed on -> logical comparison:
            (verdicts uint8[count] with bit0 = z_err, bit1 = x_err; tally int64[16]).  The literal per-trial pipeline of the reference
            (engine.py:68-122) with every stage batched on the device; used only for batches the fused OSD-0 plan left unsatisfied."""
            spz, tz, spx, tx = self.plan.sample(base_seed, begin, count)
            tally = np.zeros(_lib.TALLY_SLOTS, np.int64)
            verdict = np.zeros(count, np.uint8)
            tally[T["trials"]] = count
            for sec, (g, llrs, mask, synd, true, alpha) in enumerate(((self.graphs[0], llrs_z, masks[0], spz, tz, alpha_z), (self.graphs[1], llrs_x, masks[1], spx, tx, alpha_x))):
                det, conv, llr, iters = _lib.minsum_decode_batch(g, synd, llrs, maxIter, alpha_mode, alpha)
                failed = np.flatnonzero(conv == 0)
                if failed.size:                                                          # engine.py:96-97 / 115-116
                    det[failed] = _lib.osdw_batch(g, synd[failed], llr[failed], det[failed], osd_order)
                rows = np.stack([(mask >> np.uint64(r)) & np.uint64(1) for r in range(k)]).astype(np.int64)       # k x n logical rows
                dec = (det.astype(np.int64) @ rows.T) % 2                                # engine.py:99 / 119
                err = np.any(dec != true.astype(np.int64), axis=1)                       # engine.py:100 / 120
                verdict |= (err.astype(np.uint8) << sec)
                sfx = "zx"[sec]
                tally[T[sfx + "_err"]] = int(err.sum())
                tally[T["bp_conv_" + sfx]] = int(conv.sum())
                tally[T["osd_" + sfx]] = int(failed.size)
                tally[T["iters_" + sfx]] = int((iters.astype(np.int64) + 1).sum())
                tally[T["zero_synd_" + sfx]] = int((~synd.any(axis=1)).sum())
                tally[T["unsat_" + sfx]] = int((_lib.gf2_spmv_batch(g, det) != synd).any(axis=1).sum())
            tally[T["total_err"]] = int(np.count_nonzero(verdict))
            return verdict, tally

        def run_outcomes(self, begin, count):
            """(verdicts in trial order, tally) of [begin, begin + count): the in-order early stop needs every trial's verdict (engine.py:441-464)."""
            if not count:
                return np.zeros(0, np.uint8), np.zeros(_lib.TALLY_SLOTS, np.int64)
            local = self.plan.run_outcomes(base_seed, begin, count, self.stream.ptr)
            tally = self.plan.read(self.stream.ptr, clear=True)
            if osd_order > 0 and (tally[T["unsat_z"]] or tally[T["unsat_x"]]):
                local, tally = self.osdw_batch(begin, count)
            return local, tally

        def run(self, begin, count):
            """tally of [begin, begin + count).  The fused plan runs OSD-0, which IS the reference's OSD-w answer whenever it reproduces the syndrome
            (osd.py:27-29) -- every syndrome a circuit can produce.  So the whole range goes through at full speed (no host round trip between batches)
            and the unsatisfied counters are looked at once; only if a trial was left unsatisfied (foreign decoding matrices) is the range redone
            batch by batch with OSD-w."""
            if count:
                self.plan.run(base_seed, begin, count, self.stream.ptr)
            tally = self.plan.read(self.stream.ptr, clear=True)
            if osd_order > 0 and (tally[T["unsat_z"]] or tally[T["unsat_x"]]):
                tally = np.zeros(_lib.TALLY_SLOTS, np.int64)
                for off in range(0, count, batch):
                    nb = min(batch, count - off)
                    self.plan.run(base_seed, begin + off, nb, self.stream.ptr)
                    t = self.plan.read(self.stream.ptr, clear=True)
                    if t[T["unsat_z"]] or t[T["unsat_x"]]:
                        t = self.osdw_batch(begin + off, nb)[1]
                    tally += t
            return tally

        def close(self):
            self.plan.close()
            self.stream.close()

    workers = [Worker(dev, graphs if i == 0 else None) for i, dev in enumerate(devices)]
    W = len(workers)
    pool = ThreadPoolExecutor(max_workers=W) if W > 1 else None

    def on_workers(method, begin, count):
        """`method` of every worker on its share of [begin, begin + count) (contiguous, in worker order), concurrently: one host thread per worker
        enqueues and waits -- the calls into the library release the GIL, and a thread that blocks on a full launch queue holds up its own GPU only."""
        parts = [parallel.shard_range(count, w, W) for w in range(W)]
        if pool is None:
            return [getattr(workers[0], method)(begin + parts[0][0], parts[0][1])]
        futs = [pool.submit(getattr(workers[w], method), begin + parts[w][0], parts[w][1]) for w in range(W)]
        return [f.result() for f in futs]

    if max_trials is None:
        max_trials = num_trials if num_trials is not None else 1000000
    stop_on_errors = target_logical_errors is not None and target_logical_errors > 0
    total = np.zeros(_lib.TALLY_SLOTS, np.int64)
    done = 0
    z_errs = x_errs = total_errs = 0
    lanes = world * W                                 # plans working on a round, over all ranks
    round_size = min(batch, 1024) * lanes if stop_on_errors else max_trials
    try:
        while done < max_trials:
            if stop_on_errors and done > 0:
                # size the next round from the error rate seen so far (x1.25 head-room): big batches keep the GPUs full, but every trial
                # decoded beyond the stop point is wasted work (the count itself stays exact through the prefix cut below)
                need = (target_logical_errors - total_errs) * done / max(1, total_errs) * 1.25 if total_errs else 4.0 * done
                round_size = int(min(batch * lanes, max(256 * lanes, need)))
            this = min(round_size, max_trials - done)
            begin, count = parallel.shard_range(this, rank, world)
            if stop_on_errors:
                # the reference consumes trials in index order and stops AT the trial that brings the count to the target
                # (engine.py:441-464); per-trial verdicts in shot order + a prefix cut reproduce trials_run exactly
                res = on_workers("run_outcomes", done + begin, count)
                local = np.concatenate([r[0] for r in res])
                local_tally = np.sum([r[1] for r in res], axis=0)
                verdicts = parallel.gather_in_shot_order(local, this, device=device)
                keep = parallel.cut_at_target(verdicts, total_errs, target_logical_errors)
                head = verdicts[:keep]
                z_errs += int(np.count_nonzero(head & 1))
                x_errs += int(np.count_nonzero(head & 2))
                total_errs += int(np.count_nonzero(head))
                total += parallel.allreduce_tally(local_tally, device=device)             # diagnostics only: whole rounds
                done += keep
                if total_errs >= target_logical_errors:
                    break
            else:
                local_tally = np.sum(on_workers("run", done + begin, count), axis=0)
                total += parallel.allreduce_tally(local_tally, device=device)           # engine.py:450-457
                done += this
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
    extra["num_workers"] = W
    extra["devices"] = list(devices)
    plan = workers[0].plan
    try:
        ph, nbat = plan.phase_times()                                            # hipEvent spans of the plan's batches (an extension: not in the reference's result)
        extra["phase_ms_per_batch"] = {k: v / max(nbat, 1) for k, v in ph.items()}
    except _lib.QldpcError:
        pass
    for w in workers:
        w.close()
    if stop_on_errors:
        result = {"logical_error_rate": total_errs / max(1, done), "z_logical_error_rate": z_errs / max(1, done),
                  "x_logical_error_rate": x_errs / max(1, done), "num_trials": done, "logical_errors": total_errs}     # engine.py:466-472
    else:
        result = parallel.tally_to_result(total)
    result.update(extra)
    result["tally"] = total
    return result
