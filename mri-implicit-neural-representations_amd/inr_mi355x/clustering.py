"""Ring partition of k-space by 1-D k-means (the reference's src/clustering.py:19-135), the preprocessing
that gives the multiscale loop its radii (train_kspace_multiscale.py:73-84).

The ring statistics run on whatever device holds the k-space (3.5 M points x 40 rings of masked reductions);
the k-means itself is 40 numbers and uses the same ``sklearn.cluster.KMeans(init="random", n_init=10,
max_iter=200, random_state=42)`` call as the reference, on the host.  Nothing is plotted.
"""
from math import sqrt
from typing import Tuple

import numpy as np
import torch


def _complex_abs(x: torch.Tensor) -> torch.Tensor:
    """fastmri.complex_abs: sqrt(re^2 + im^2) over the last axis (size 2)."""
    return torch.sqrt((x ** 2).sum(dim=-1))


def ring_bounds(no_steps: int):
    """clustering.py:48-57: ring i covers [sqrt2*i/n, sqrt2*(i+1)/n], both ends included."""
    out = []
    for i in range(no_steps):
        r0 = 0 if i == 0 else sqrt(2) * i / no_steps
        r1 = sqrt(2) if i == no_steps - 1 else sqrt(2) * (i + 1) / no_steps
        out.append((r0, r1))
    return out


def partition_kspace(img: torch.Tensor, kcoords: torch.Tensor, no_steps: int = 40, no_parts: int = 4):
    """clustering.py:19-89.  img [C,H,W,2], kcoords [C,H,W,3] (coil, y, x).  Returns (labels of the no_steps
    initial rings, radii [no_parts+1] separating the final partitions; the last radius is 5 = 'everything')."""
    from sklearn.cluster import KMeans
    dist = torch.sqrt(kcoords[..., 1] ** 2 + kcoords[..., 2] ** 2)
    logmag = torch.log(_complex_abs(img))
    means = []
    for r0, r1 in ring_bounds(no_steps):
        sel = (dist >= r0) & (dist <= r1)
        means.append(float(logmag[sel].max()))  # clustering.py:60-61 (named means, is the max)
    means = np.array(means).reshape(-1, 1)
    kmeans = KMeans(init="random", n_clusters=no_parts, n_init=10, max_iter=200, random_state=42)
    kmeans.fit(means)
    labels = kmeans.labels_
    # Walking outwards, every cluster claims as many of the initial rings as carry its label, in the order in which
    # the clusters are first met (clustering.py:72-84): boundary j = sqrt(2) * (rings claimed by the first j clusters)
    # / no_steps.  (Rings of one cluster are counted together even if another cluster's ring sits between them.)
    first_met, ring_count = [], {}
    for lab in labels.tolist():
        if lab not in ring_count:
            first_met.append(lab)
            ring_count[lab] = 0
        ring_count[lab] += 1
    claimed = np.cumsum([ring_count[lab] / len(labels) for lab in first_met])
    radii = np.concatenate(([0.0], sqrt(2) * claimed))
    radii[no_parts] = 5  # the outermost partition is "everything beyond" (clustering.py:82)
    return labels, radii


def partition_and_stats(img: torch.Tensor, kcoords: torch.Tensor, no_steps: int = 40, no_parts: int = 4,
                        stat: str = "max") -> Tuple[torch.Tensor, np.ndarray]:
    """clustering.py:91-135: per final partition the max (or min) of |component| of the k-space inside it."""
    _, radii = partition_kspace(img, kcoords, no_steps, no_parts)
    dist = torch.sqrt(kcoords[..., 1] ** 2 + kcoords[..., 2] ** 2)
    stats = []
    for i in range(len(radii) - 1):
        sel = (dist >= radii[i]) & (dist <= radii[i + 1])
        a = torch.abs(img[sel])
        stats.append(a.min() if stat == "min" else a.max())
    return torch.stack(stats), radii
