"""CPU: the N>1 prompt-sharding path with world_size 2 over gloo (no GPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plangen_amd.dist import all_gather_rows, broadcast_prompts, gather_rows, shard_range


def test_shard_range_partitions():
    for n in (1, 2, 7, 64, 255):
        for ws in (1, 2, 4, 8):
            spans = [shard_range(n, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, ws, port, B, L, T, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 1000, (2 * B, L), generator=g).int()
    mask = torch.ones((2 * B, L + T), dtype=torch.int32)
    my_ids, my_mask, lo, hi, nB = broadcast_prompts(ids if rank == 0 else None, mask if rank == 0 else None, "cpu")
    assert nB == B and my_ids.shape == (2 * (hi - lo), L) and my_mask.shape == (2 * (hi - lo), L + T)
    assert torch.equal(my_ids, ids[2 * lo:2 * hi])            # CFG pairs stay together
    # stand-in for generation: tokens derived from the cond row only
    toks = my_ids[0::2, :T].clone()
    allt = gather_rows(toks, B)                                # a gather to rank 0, not an all-gather
    if rank == 0:
        assert torch.equal(allt, ids[0::2, :T])
    else:
        assert allt is None
    imgs = gather_rows(my_ids[0::2, :3].float().view(-1, 3, 1, 1), B)
    if rank == 0:
        assert torch.equal(imgs.view(B, 3), ids[0::2, :3].float())
    assert torch.equal(all_gather_rows(toks, B), ids[0::2, :T])
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_gather_world2():
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 5, 6, 4, None), nprocs=2, join=True)


def _bench(*argv, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_bench_launcher_spawns_ranks_itself():
    """`python bench.py --gpus N` with no WORLD_SIZE: the parent starts N rank processes (env rendezvous on
    127.0.0.1), relays ONE JSON line from rank 0 and exits 0 (dry run: gloo, no GPU)."""
    rc, out, err = _bench("--gpus", "2", "--launch-check", "--global-batch", "6")
    assert rc == 0, err
    assert out == {"launch_check": "ok", "world": 2, "global_images": 6, "images_rank0": 3, "backend": "gloo"}
    rc, out, err = _bench("--gpus", "3", "--launch-check", "--batch", "4")
    assert rc == 0 and out["world"] == 3 and out["global_images"] == 12, err


def test_bench_refuses_mislabelled_world_size():
    """Launched by torchrun with a world size that is not --gpus: non-zero exit, no JSON line."""
    rc, out, err = _bench("--gpus", "4", "--launch-check", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0",
                                                                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    assert rc != 0 and out is None and "WORLD_SIZE" in err


def test_bench_launcher_config4_dry_run():
    """BASELINE configs[3]: 256 images over 8 ranks (32 per rank) through the launcher's own spawn path (gloo dry run)."""
    rc, out, err = _bench("--gpus", "8", "--launch-check", "--global-batch", "256")
    assert rc == 0, err
    assert out == {"launch_check": "ok", "world": 8, "global_images": 256, "images_rank0": 32, "backend": "gloo"}


def test_bench_launcher_fails_fast_when_a_rank_dies():
    """Rank 1 exits before the first collective: the launcher must notice (it polls every child), terminate the surviving
    ranks -- which would otherwise sit in the all-reduce until the collective timeout -- and return non-zero promptly."""
    import time
    t0 = time.time()
    rc, out, err = _bench("--gpus", "3", "--launch-check", "--batch", "2", env={"PG_TEST_DIE_RANK": "1", "PG_DIST_TIMEOUT_S": "120"})
    dt = time.time() - t0
    assert rc != 0 and out is None
    assert "rank 1 exited with rc 7" in err and "terminating the other ranks" in err, err
    assert dt < 60, dt                                     # far below the 120 s collective timeout


def test_force_collectives_runs_them_in_a_one_rank_group():
    """``force_collectives=True`` does not short-circuit at world 1 (the single-GPU RCCL self-test uses it; here over gloo)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "rccl_one_rank.py"), "gloo"], env=e, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0 and '"rccl_one_rank": "ok"' in p.stdout, p.stderr[-2000:]


def _wait_pids(d, n, timeout=120):
    import time
    t0 = time.time()
    while time.time() - t0 < timeout:
        fs = [f for f in os.listdir(d) if f.endswith(".pid")]
        if len(fs) == n:
            return [int(open(os.path.join(d, f)).read()) for f in fs]
        time.sleep(0.2)
    raise AssertionError("ranks did not start")


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    try:                                            # a zombie still answers signal 0
        return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
    except FileNotFoundError:
        return False


def _launcher_signal_case(sig, tmp_path):
    import signal
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, PG_TEST_HANG_S="600", PG_TEST_PID_DIR=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check", "--batch", "2"], env=e,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        pids = _wait_pids(str(tmp_path), 2)
        assert all(_alive(q) for q in pids)
        p.send_signal(sig)
        if sig == signal.SIGKILL:
            p.wait(timeout=30)
        else:
            _, err = p.communicate(timeout=60)
            assert p.returncode == 130 and "launcher interrupted" in err, (p.returncode, err)
        t0 = time.time()
        while any(_alive(q) for q in pids) and time.time() - t0 < 30:
            time.sleep(0.2)
        assert not any(_alive(q) for q in pids), "rank processes survived the launcher"
    finally:
        if p.poll() is None:
            p.kill()
        for f in os.listdir(str(tmp_path)):
            try:
                os.kill(int(open(os.path.join(str(tmp_path), f)).read()), 9)      # exact PIDs the ranks wrote themselves
            except (ProcessLookupError, ValueError):
                pass


def test_launcher_sigterm_takes_the_ranks_down(tmp_path):
    """the ranks run in their own sessions; SIGTERM to the launcher (a `timeout`, a driver kill) must still end them."""
    import signal
    _launcher_signal_case(signal.SIGTERM, tmp_path)


def test_launcher_sighup_takes_the_ranks_down(tmp_path):
    import signal
    _launcher_signal_case(signal.SIGHUP, tmp_path)


def test_launcher_sigkill_takes_the_ranks_down_through_pdeathsig(tmp_path):
    """A SIGKILLed launcher cannot run any handler: PR_SET_PDEATHSIG in every rank covers it."""
    import signal
    _launcher_signal_case(signal.SIGKILL, tmp_path)


# ---------------------------------------------------------------------------------------------- rank placement (plangen_amd/affinity.py)
def _fake_sysfs(root, gpu_numa, node_cpus, cpu_nodes=2):
    """A sysfs tree with ``cpu_nodes`` KFD CPU nodes followed by one KFD GPU node per entry of gpu_numa (render minors 128..)."""
    kfd = root / "class" / "kfd" / "kfd" / "topology" / "nodes"
    for n in range(cpu_nodes):
        (kfd / str(n)).mkdir(parents=True)
        (kfd / str(n) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
    for i, numa in enumerate(gpu_numa):
        d = kfd / str(cpu_nodes + i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + i}\n")
        dev = root / "class" / "drm" / f"renderD{128 + i}" / "device"
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text(f"{numa}\n")
    for k, cl in node_cpus.items():
        nd = root / "devices" / "system" / "node" / f"node{k}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cl + "\n")
    return str(root)


def test_affinity_cpulist_round_trip():
    from plangen_amd.affinity import format_cpulist, parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert format_cpulist([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"
    assert parse_cpulist("") == [] and format_cpulist([]) == ""


def test_affinity_plan_8_gpus_2_sockets(tmp_path):
    """The config-4 box: 8 GPUs, 4 per socket, 2 x 64 cores with SMT siblings 128-255: every rank gets a disjoint slice of ITS GPU's node."""
    from plangen_amd.affinity import gpu_numa_nodes, plan
    sysfs = _fake_sysfs(tmp_path, [0, 0, 0, 0, 1, 1, 1, 1], {0: "0-63,128-191", 1: "64-127,192-255"})
    allowed = list(range(256))
    assert gpu_numa_nodes(sysfs, env={}) == [0, 0, 0, 0, 1, 1, 1, 1]
    plans = [plan(r, 8, sysfs, allowed, env={}) for r in range(8)]
    sets = [set(p["cpus"]) for p in plans]
    assert all(len(s) == 32 for s in sets)
    assert all(sets[a].isdisjoint(sets[b]) for a in range(8) for b in range(a + 1, 8))
    node0, node1 = set(range(0, 64)) | set(range(128, 192)), set(range(64, 128)) | set(range(192, 256))
    assert all(sets[r] <= node0 for r in range(4)) and all(sets[r] <= node1 for r in range(4, 8))
    assert set().union(*sets[:4]) == node0 and set().union(*sets[4:]) == node1
    assert [p["numa_node"] for p in plans] == [0, 0, 0, 0, 1, 1, 1, 1]
    # a container cpuset narrower than the node is respected: only allowed CPUs are handed out
    p = plan(5, 8, sysfs, allowed=list(range(64, 96)), env={})
    assert set(p["cpus"]) <= set(range(64, 96)) and len(p["cpus"]) == 8
    # ... and a cpuset with nothing on the GPU's node leaves the affinity alone instead of pinning to the wrong socket
    assert plan(5, 8, sysfs, allowed=list(range(0, 8)), env={})["cpus"] is None


def test_affinity_plan_visible_devices_and_degenerate_boxes(tmp_path):
    from plangen_amd.affinity import gpu_numa_nodes, plan
    sysfs = _fake_sysfs(tmp_path / "a", [0, 0, 1, 1], {0: "0-15", 1: "16-31"})
    # HIP ordinal 0 is physical GPU 2 under HIP_VISIBLE_DEVICES=2,3
    assert gpu_numa_nodes(sysfs, env={"HIP_VISIBLE_DEVICES": "2,3"}) == [1, 1]
    p = plan(0, 2, sysfs, list(range(32)), env={"HIP_VISIBLE_DEVICES": "2,3"})
    assert p["numa_node"] == 1 and p["cpus"] == list(range(16, 24))
    # one GPU on a box whose firmware reports no affinity (numa_node -1, as on the 1-GPU bench boxes): nothing is pinned
    one = _fake_sysfs(tmp_path / "b", [-1], {0: "0-7"}, cpu_nodes=1)
    p = plan(0, 1, one, list(range(8)), env={})
    assert p["cpus"] is None and "numa_node -1" in p["reason"]
    # one GPU with a node: the whole node
    one = _fake_sysfs(tmp_path / "c", [0], {0: "0-7"}, cpu_nodes=1)
    assert plan(0, 1, one, list(range(8)), env={})["cpus"] == list(range(8))
    # no KFD topology at all (this build container): reported, nothing pinned
    p = plan(0, 1, str(tmp_path / "missing"), list(range(8)), env={})
    assert p["cpus"] is None and "no KFD topology" in p["reason"]
    # more ranks than GPUs
    assert plan(3, 4, one, list(range(8)), env={})["cpus"] is None


def test_affinity_apply_pins_this_process_and_reports(tmp_path):
    """apply() in a child process (the test runner's own mask must not change): the child's mask becomes the planned slice."""
    import json
    import subprocess
    import sys
    mine = sorted(os.sched_getaffinity(0))
    if len(mine) < 2:
        import pytest
        pytest.skip("needs >= 2 allowed CPUs")
    sysfs = _fake_sysfs(tmp_path, [0, 0], {0: ",".join(str(c) for c in mine)}, cpu_nodes=1)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, os, sys; sys.path.insert(0, %r); from plangen_amd.affinity import apply; "
            "p = apply(1, 2, %r); p['mask'] = sorted(os.sched_getaffinity(0)); print(json.dumps(p))" % (root, sysfs))
    env = {k: v for k, v in os.environ.items() if not k.endswith("_VISIBLE_DEVICES")}       # this container hides every GPU (HIP_VISIBLE_DEVICES='')
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=env)
    assert out.returncode == 0, out.stderr
    p = json.loads(out.stdout.strip().splitlines()[-1])
    half = len(mine) // 2
    assert p["applied"] and p["mask"] == mine[half:] and p["numa_node"] == 0
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=dict(env, PG_NO_AFFINITY="1"))
    p = json.loads(out.stdout.strip().splitlines()[-1])
    assert not p["applied"] and p["mask"] == mine
