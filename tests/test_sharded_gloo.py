"""world_size-2 test of the N>1 path on CPU (gloo): each rank solves its shard of a
global batch and the stop-flag all-reduce makes every rank agree on the global
convergence record.  The GPU solver is replaced by the CPU oracle here (tests may
use the oracle; the product's sharded driver itself never does)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, n, m, limit, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from cppnumericalsolvers_amd import sharded
    from cppnumericalsolvers_amd.engine import synthetic_x0_host

    stop = oracle_lib.parity_stop()
    stop.num_iterations = limit

    class OracleSolver:
        """Stands in for BatchedLbfgs on a box without a GPU: same minimize() contract (tensors in, tensors and the
        40-byte progress records out), the CPU oracle underneath.  Everything else — shard ranges, the device-side
        view of the progress records, the counts and the all-reduce — is the product's ShardedLbfgs."""

        def minimize(self, objective, x0):
            x, f, g, p = oracle_lib.minimize_batch(objective, x0.numpy(), m=m, stop=stop, nthreads=2)
            return (torch.from_numpy(x), torch.from_numpy(f), torch.from_numpy(g),
                    torch.from_numpy(p.view(np.uint8).copy()))

    driver = sharded.ShardedLbfgs(OracleSolver(), rank=rank, world_size=world)
    (lo, hi), (x, f, g, prog), flag = driver.minimize_global(
        "rosenbrock", B, lambda first, count: torch.from_numpy(synthetic_x0_host(count, n, "std", first_problem=first)))
    x, f = x.numpy(), f.numpy()
    p = prog.numpy().view(oracle_lib.PROGRESS_DTYPE)
    np.savez(os.path.join(tmpdir, "rank%d.npz" % rank), x=x, f=f, lo=lo, hi=hi, total=flag.total,
             unconverged=flag.unconverged, iterations=flag.iterations, ok=flag.all_converged,
             local_bad=int((p["status"] <= 1).sum()), local_it=int(p["num_iterations"].sum()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("limit,expect_all", [(10000, True), (40, False)])
def test_two_rank_sharded_solve_and_stop_flag(tmp_path, limit, expect_all):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from cppnumericalsolvers_amd.engine import synthetic_x0_host
    B, n, m, world = 37, 16, 5, 2     # ragged split: 18 + 19
    mp.spawn(_worker, args=(world, _free_port(), B, n, m, limit, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(world)]
    # both ranks hold the same global record
    for key in ("total", "unconverged", "iterations", "ok"):
        assert r[0][key] == r[1][key], key
    assert int(r[0]["total"]) == B
    assert int(r[0]["unconverged"]) == int(r[0]["local_bad"]) + int(r[1]["local_bad"])
    assert int(r[0]["iterations"]) == int(r[0]["local_it"]) + int(r[1]["local_it"])
    assert bool(r[0]["ok"]) == expect_all
    # the sharded result equals the unsharded solve of the whole batch
    stop = oracle_lib.parity_stop()
    stop.num_iterations = limit
    x0 = synthetic_x0_host(B, n, "std")
    xg, fg, _, _ = oracle_lib.minimize_batch("rosenbrock", x0, m=m, stop=stop)
    xs = np.concatenate([r[0]["x"], r[1]["x"]])
    assert (int(r[0]["lo"]), int(r[0]["hi"]), int(r[1]["lo"]), int(r[1]["hi"])) == (0, 18, 18, 37)
    np.testing.assert_array_equal(xs, xg)


def _al_worker(rank, world, port, B, n, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import auglag_lib as al
    from cppnumericalsolvers_amd import sharded
    from cppnumericalsolvers_amd.engine import synthetic_x0_host

    lo, hi = sharded.shard_range(B, rank, world)
    x0 = synthetic_x0_host(hi - lo, n, "std", first_problem=lo)
    r = al.oracle_minimize(al.quadratic_simplex_problem(n), x0, config=al.default_config(outer_num_iterations=30),
                           nthreads=2)
    p = r["progress"]
    flag = sharded.allreduce_flag(sharded.local_counts(p["status"], p["num_iterations"]))
    np.savez(os.path.join(tmpdir, "al_rank%d.npz" % rank), x=r["x"], lam=r["lambda"], total=flag.total,
             unconverged=flag.unconverged, iterations=flag.iterations,
             local_bad=int((p["status"] <= 1).sum()), local_it=int(p["num_iterations"].sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_augmented_lagrangian(tmp_path):
    """The constrained path shards like the unconstrained one: outer loops of different shards run independently
    (their outer-iteration counts differ) and only the 3-word record is exchanged."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import auglag_lib as al
    from cppnumericalsolvers_amd.engine import synthetic_x0_host
    B, n, world = 21, 6, 2
    mp.spawn(_al_worker, args=(world, _free_port(), B, n, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "al_rank%d.npz" % k)) for k in range(world)]
    for key in ("total", "unconverged", "iterations"):
        assert r[0][key] == r[1][key], key
    assert int(r[0]["total"]) == B
    assert int(r[0]["unconverged"]) == int(r[0]["local_bad"]) + int(r[1]["local_bad"])
    assert int(r[0]["iterations"]) == int(r[0]["local_it"]) + int(r[1]["local_it"])
    whole = al.oracle_minimize(al.quadratic_simplex_problem(n), synthetic_x0_host(B, n, "std"),
                               config=al.default_config(outer_num_iterations=30))
    np.testing.assert_array_equal(np.concatenate([r[0]["x"], r[1]["x"]]), whole["x"])
    np.testing.assert_array_equal(np.concatenate([r[0]["lam"], r[1]["lam"]]), whole["lambda"])


def test_al_progress_fields_device_view():
    from cppnumericalsolvers_amd import capi, sharded
    rec = np.zeros(4, dtype=capi.AL_PROGRESS_DTYPE)
    rec["status"] = [6, 1, 6, 0]
    rec["num_iterations"] = [3, 41, 5, 7]
    st, it = sharded.al_progress_fields_device(torch.from_numpy(rec.view(np.uint8).copy()))
    assert st.tolist() == [6, 1, 6, 0] and it.tolist() == [3, 41, 5, 7]


def test_allreduce_flag_without_process_group():
    from cppnumericalsolvers_amd import sharded
    f = sharded.allreduce_flag(sharded.local_counts(np.array([2, 4, 1, 3]), np.array([5, 6, 7, 8])))
    assert (f.total, f.unconverged, f.iterations, f.all_converged) == (4, 1, 26, False)
    t = sharded.local_counts(torch.tensor([2, 4, 4], dtype=torch.int32), torch.tensor([1, 2, 3], dtype=torch.int32))
    f = sharded.allreduce_flag(t)
    assert (f.total, f.unconverged, f.iterations, f.all_converged) == (3, 0, 6, True)


def test_progress_fields_device_view():
    from cppnumericalsolvers_amd import capi, sharded
    rec = np.zeros(5, dtype=capi.PROGRESS_DTYPE)
    rec["status"] = [2, 4, 1, 3, 4]
    rec["num_iterations"] = [10, 20, 30, 40, 50]
    rec["nfev"] = [11, 21, 31, 41, 51]
    rec["sum_k"] = [1, 2, 3, 4, 5]
    buf = torch.from_numpy(rec.view(np.uint8).copy())
    st, it, nf, sk = sharded.progress_fields_device(buf)
    assert st.tolist() == [2, 4, 1, 3, 4] and it.tolist() == [10, 20, 30, 40, 50]
    assert nf.tolist() == [11, 21, 31, 41, 51] and sk.tolist() == [1, 2, 3, 4, 5]
