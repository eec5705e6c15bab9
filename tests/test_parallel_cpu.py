"""N > 1 path on CPU: world_size-2 (and 3) gloo processes shard the shot range, each tallies its shard (here with the
oracle as the stand-in compute, since there is no GPU), one all-reduce; the result must equal the single-process tally."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    import qldpc_amd  # noqa: F401
    from qldpc_amd import parallel
    from qldpc_amd.data import load_code
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = load_code("bb72")

    def local(begin, count):
        return orc.cc_sample_decode_tally(c["Hx_indptr"], c["Hx_indices"], c["n"], c["Lx"], 0.04, 99, begin, count, max_iter=20, threads=1)
    t = parallel.run_sharded(total, local)
    np.save(os.path.join(out_dir, f"tally_{rank}.npy"), t)
    # per-trial verdicts gathered back into shot order (the in-order early stop of engine.py:441-464)
    for round_total in (0, 1, world - 1, 257):
        begin, count = parallel.shard_range(round_total, rank, world)
        mine = ((np.arange(begin, begin + count) * 7919) % 11 == 0).astype(np.uint8) * (1 + (np.arange(begin, begin + count) % 3 == 0))
        np.save(os.path.join(out_dir, f"verdicts_{round_total}_{rank}.npy"), parallel.gather_in_shot_order(mine, round_total))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_tally_equals_single_process(tmp_path, oracle, world):
    import torch.multiprocessing as mp
    from qldpc_amd.data import load_code
    total = 4001                                   # not divisible by the world size
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    c = load_code("bb72")
    ref = oracle.cc_sample_decode_tally(c["Hx_indptr"], c["Hx_indices"], c["n"], c["Lx"], 0.04, 99, 0, total, max_iter=20, threads=1)
    for r in range(world):
        t = np.load(os.path.join(str(tmp_path), f"tally_{r}.npy"))
        assert np.array_equal(t, ref), (r, t.tolist(), ref.tolist())
    assert ref[0] == total and ref[3] > 0
    for round_total in (0, 1, world - 1, 257):
        idx = np.arange(round_total)
        want = ((idx * 7919) % 11 == 0).astype(np.uint8) * (1 + (idx % 3 == 0))
        for r in range(world):
            assert np.array_equal(np.load(os.path.join(str(tmp_path), f"verdicts_{round_total}_{r}.npy")), want)


def test_cut_at_target():
    import qldpc_amd  # noqa: F401
    from qldpc_amd.parallel import cut_at_target
    v = np.array([0, 1, 0, 0, 2, 3, 0, 1], np.uint8)
    assert cut_at_target(v, 0, 1) == 2 and cut_at_target(v, 0, 2) == 5 and cut_at_target(v, 0, 4) == 8
    assert cut_at_target(v, 0, 5) == 8 and cut_at_target(v, 3, 4) == 2 and cut_at_target(v, 4, 4) == 0
    assert cut_at_target(np.zeros(0, np.uint8), 0, 1) == 0


def test_shard_range_properties():
    import qldpc_amd  # noqa: F401
    from qldpc_amd.parallel import shard_range, tally_to_result
    for total in (0, 1, 7, 1000, 10 ** 7 + 3):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for (b0, c0), (b1, _) in zip(parts, parts[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 3, 2)
    t = np.zeros(16, np.int64); t[0] = 200; t[1] = 3; t[3] = 3
    r = tally_to_result(t)
    assert r["num_trials"] == 200 and r["logical_errors"] == 3 and abs(r["logical_error_rate"] - 0.015) < 1e-12
