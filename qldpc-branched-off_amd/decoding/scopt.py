"""SCOPT beta estimator (interface of the reference's src/decoding/scopt.py).

Same procedure as the alpha estimators, applied to the decoder OUTPUT: decode `trials` sampled error patterns with early exit,
histogram the final posteriors by the true bit value, fit log(f1/f0) = beta * LLR.  The decode of all trials is one batched call
of the min-sum kernels; the posteriors stay on the device and only the 2 x `bins` counters return.
"""
import numpy as np

from .. import _lib
from ._fit import class_densities, draw_errors, graph_of, save_fit_plot, slope_and_r2


def estimate_scopt_beta(code, error_rate, trials=10000, bins=50, alpha=1.0, alpha_mode="dynamical", maxIter=50, damping=1.0, clip_llr=20.0,
                        rng=None, plot_dir=None, plot_prefix=None, llrs=None):
    """scopt.py:8-177 -> (beta, r2)."""
    if error_rate <= 0 or error_rate >= 0.5:
        raise ValueError("error_rate must be in (0, 0.5)")
    if rng is None:
        rng = np.random.default_rng()
    graph, (_, n) = graph_of(code)
    if maxIter <= 0:
        raise ValueError("maxIter must be > 0")
    if alpha_mode not in {"dynamical", "alvarado", "alvarado-autoregressive"}:
        raise ValueError(f"Unsupported alpha_mode: {alpha_mode}")
    if alpha_mode == "alvarado-autoregressive":
        alpha = np.asarray(alpha, dtype=np.float64)
        if alpha.ndim != 1 or alpha.size == 0:
            raise ValueError("alpha must be a non-empty 1D sequence for alvarado-autoregressive")
    elif alpha_mode == "alvarado":
        alpha = float(alpha)
    if trials <= 0:
        raise ValueError("Insufficient samples for beta estimation")
    # scopt.py:93-94 uses float(alpha) in "alvarado" mode without the wrappers' alpha > 0 check: pass it through as a constant
    mode_for_device = None if alpha_mode == "alvarado" else alpha_mode
    const_alpha = alpha if alpha_mode != "dynamical" else 1.0
    if alpha_mode == "alvarado" and alpha == 0:
        raise ValueError("alpha = 0 is not a usable constant normalisation factor")
    stats = _lib.MessageStats(graph, draw_errors(rng, trials, n, error_rate), llrs, _lib.STATS_POSTERIOR, maxIter, alpha_mode=mode_for_device,
                              alpha=const_alpha, damping=damping, clip_llr=clip_llr)
    try:
        f0, f1, edges = class_densities(stats, bins, "beta")
    finally:
        stats.close()
    centres = (edges[:-1] + edges[1:]) / 2.0
    both = (f0 > 0) & (f1 > 0)
    log_ratio = np.log(f1[both] / f0[both])                     # note the orientation: f1 over f0 (scopt.py:157)
    x = centres[both]
    beta, r2 = slope_and_r2(x, log_ratio)
    if plot_dir is not None:
        save_fit_plot(f"{plot_dir}/{plot_prefix or f'beta_p{error_rate:.6g}'}_beta_fit.png", x, log_ratio, beta, r2, "LLR", "log(f1/f0)",
                      f"SCOPT beta fit (p={error_rate:.6g})", "#64B791")
    return beta, r2
