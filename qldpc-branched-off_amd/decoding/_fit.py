"""Shared tail of the alpha / beta estimators: class histograms from the device -> one-parameter line through the origin.

The reference builds both histograms with np.histogram(..., range=(lo, hi), density=True) over the concatenated samples of all
trials and fits log(f0/f1) = slope * bin_centre with scipy's curve_fit (src/decoding/alpha.py:26-66, scopt.py:139-166).  Here the
samples never leave the GPU: `MessageStats` returns the finite range and the integer counts per bin, and the density
normalisation below is numpy's own (counts / bin width / total count).
"""
import numpy as np

from .. import _lib


def draw_errors(rng, trials, n, error_rate):
    """The `trials` successive `rng.random(n) < error_rate` draws of the reference loops, taken in one call
    (a numpy Generator fills a (trials, n) request in the same stream order)."""
    return (np.asarray(rng.random((int(trials), int(n)))) < error_rate).astype(np.int8)


def graph_of(code):
    """Parity-check matrix (dense, scipy CSR, ...) or an existing `_lib.Graph` -> (Graph, (m, n))."""
    if isinstance(code, _lib.Graph):
        return code, (code.m, code.n)
    indptr, indices, shape = _lib.canonical_csr(code)
    return _lib.graph_for(indptr, indices, shape[1]), shape


def class_densities(stats, bins, what):
    """-> (density of class 0, density of class 1, bin edges); raises the reference's ValueErrors for empty classes."""
    if stats.finite[0] == 0 or stats.finite[1] == 0:
        raise ValueError(f"No finite samples for {what} estimation")
    edges = np.histogram_bin_edges(np.zeros(0), bins=bins, range=stats.range)
    widths = np.array(np.diff(edges), float)
    dens = [counts / widths / counts.sum() for counts in stats.histogram(edges)]
    return dens[0], dens[1], edges


def slope_and_r2(x, y):
    """Least-squares slope of y = slope * x with scipy's curve_fit (as the reference) and the R^2 it reports."""
    from scipy.optimize import curve_fit

    def through_origin(t, slope):
        return slope * t

    popt, _ = curve_fit(through_origin, x, y)
    slope = popt[0]
    resid = y - through_origin(x, slope)
    spread = np.sum((y - np.mean(y)) ** 2)
    return slope, 1.0 - (np.sum(resid ** 2) / spread if spread > 0 else np.nan)


def save_fit_plot(path, x, y, slope, r2, xlabel, ylabel, title, colour):
    """Optional diagnostic figure (only when the caller passes plot_dir); matplotlib is imported lazily."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots(figsize=(6, 4))
    ax.scatter(x, y, s=10, alpha=0.7, label="samples")
    ax.plot(x, slope * x, color=colour, label=f"fit (R^2={r2:.3f})")
    ax.set_xlabel(xlabel)
    ax.set_ylabel(ylabel)
    ax.set_title(title)
    ax.grid(True, ls="-", alpha=0.4)
    ax.legend()
    fig.tight_layout()
    fig.savefig(path, dpi=300)
    plt.close(fig)
