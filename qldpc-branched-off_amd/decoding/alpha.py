"""Alvarado normalisation-factor estimators (interface of the reference's src/decoding/alpha.py).

Each estimator is: draw `trials` error patterns, run the decoder for k iterations, take the UNSCALED check-to-variable
messages of the next pass, histogram them by the true value of the bit they point at, fit log(f0/f1) = alpha * lambda.
The reference does this with a Python loop per trial (and per edge in the autoregressive variant); here one
`qldpc_msgstats_create` call does all trials of one fit on the GPU and only 2 x `bins` counters come back.
"""
import numpy as np

from .. import _lib
from ._fit import class_densities, draw_errors, graph_of, save_fit_plot, slope_and_r2


def _alpha_from_stats(stats, bins, plot_path=None, title=None):
    """alpha.py:26-81 on device histograms -> (alpha, r2)."""
    f0, f1, edges = class_densities(stats, bins, "alpha")
    centres = (edges[:-1] + edges[1:]) / 2.0
    both = (f0 > 0) & (f1 > 0)
    if not np.any(both):
        raise ValueError("No overlapping histogram bins for alpha estimation")
    lam = centres[both]
    log_ratio = np.log(f0[both] / f1[both])
    alpha, r2 = slope_and_r2(lam, log_ratio)
    if plot_path is not None:
        save_fit_plot(plot_path, lam, log_ratio, alpha, r2, "Lambda", "log(f0/f1)", title or "Alpha estimation linear fit", "#DBA142")
    return alpha, r2


def _check_rate(error_rate):
    if error_rate <= 0 or error_rate >= 0.5:
        raise ValueError("error_rate must be in (0, 0.5)")


def estimate_alpha_alvarado(code, error_rate, trials=5000, bins=50, rng=None, plot_dir=None, plot_prefix=None, llrs=None):
    """One-iteration estimate (alpha.py:84-159): messages of the first check pass at alpha = 1 -> (alpha, r2)."""
    _check_rate(error_rate)
    if rng is None:
        rng = np.random.default_rng()
    graph, (_, n) = graph_of(code)
    if trials <= 0:
        raise ValueError("Insufficient samples for alpha estimation")
    stats = _lib.MessageStats(graph, draw_errors(rng, trials, n, error_rate), llrs, _lib.STATS_CHECK_MESSAGES, 0)
    try:
        path = None if plot_dir is None else f"{plot_dir}/{plot_prefix or f'alvarado_p{error_rate:.6g}'}_alpha_fit.png"
        return _alpha_from_stats(stats, bins, path, f"Alvarado alpha fit (p={error_rate:.6g})")
    finally:
        stats.close()


def estimate_alpha_alvarado_autoregressive(code, error_rate, maxIter, trials=5000, bins=50, damping=1.0, clip_llr=20.0, rng=None,
                                           plot_dir=None, plot_prefix=None, llrs=None):
    """Per-iteration sequence (alpha.py:162-276): alpha_k is fitted on the unscaled messages of iteration k after the decoder
    state has been advanced with alpha_0..alpha_{k-1}; fresh error patterns per k -> (alphas f64[maxIter], r2s f64[maxIter])."""
    _check_rate(error_rate)
    if maxIter <= 0:
        raise ValueError("maxIter must be > 0")
    if rng is None:
        rng = np.random.default_rng()
    graph, (_, n) = graph_of(code)
    if trials <= 0:
        raise ValueError("Insufficient samples for alpha estimation")
    alphas, r2s = [], []
    for k in range(maxIter):
        errors = draw_errors(rng, trials, n, error_rate)
        mode, seq = ("alvarado-autoregressive", np.asarray(alphas, dtype=np.float64)) if k else ("dynamical", 1.0)   # k = 0: no previous pass
        stats = _lib.MessageStats(graph, errors, llrs, _lib.STATS_CHECK_MESSAGES, k, alpha_mode=mode, alpha=seq, damping=damping, clip_llr=clip_llr)
        try:
            path = None if plot_dir is None else f"{plot_dir}/{plot_prefix or f'autoregressive_p{error_rate:.6g}'}_iter{k + 1}_alpha_fit.png"
            a, r2 = _alpha_from_stats(stats, bins, path, f"Autoregressive alpha fit (p={error_rate:.6g}, iter={k + 1})")
        finally:
            stats.close()
        alphas.append(float(a))
        r2s.append(float(r2))
    return np.asarray(alphas, dtype=np.float64), np.asarray(r2s, dtype=np.float64)
