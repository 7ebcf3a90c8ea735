"""Synthetic two-class data for the user-objective example (the reference's example uses the Iris versicolor /
virginica split, which is not vendored here): two Gaussian blobs in d dimensions, labels +/- 1."""
import numpy as np


def two_blobs(N=100, d=4, seed=7, separation=2.0):
    rng = np.random.default_rng(seed)
    y = np.where(np.arange(N) % 2 == 0, 1.0, -1.0)
    centre = np.linspace(0.5, 1.5, d) * separation
    X = rng.normal(size=(N, d)) + np.outer(y, centre) * 0.5
    return np.ascontiguousarray(X), y


def params(X, y, C=1.0):
    return np.concatenate([[float(X.shape[0]), float(X.shape[1]), float(C)], X.ravel(), y])
