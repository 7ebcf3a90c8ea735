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


def dual_params(X, y):
    """Parameter blob of the dual-SVM user objective (examples/user_objective_svm_dual/svm_dual.hpp): n, then the
    kernel-with-labels matrix Q = (X X^T) .* (y y^T) of the reference's svm_dual_lbfgsb.cc:46-50, accumulated feature by
    feature so that Q is symmetric to the bit (every term X_ik X_jk is the same product in both triangles)."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    G = np.outer(X[:, 0], X[:, 0])
    for k in range(1, X.shape[1]):
        G = G + np.outer(X[:, k], X[:, k])
    Q = G * np.outer(y, y)
    assert np.array_equal(Q, Q.T)
    return np.concatenate([[float(X.shape[0])], Q.ravel()]), Q


def standardised_blobs(N=100, d=4, seed=7, separation=1.2):
    """z-score-centred two-class data (the reference example standardises the Iris features the same way)."""
    X, y = two_blobs(N, d, seed, separation)
    X = (X - X.mean(axis=0)) / X.std(axis=0)
    return np.ascontiguousarray(X), y
