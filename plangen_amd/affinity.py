"""Host-side placement of a rank process: pin it to the CPUs of the NUMA node its GPU hangs off.

One process per GPU enqueues ~175 short dependent launches per decode step (4 200 launches per 24-layer step at 32 images per GPU the loop is
launch-bound: the enqueue thread must stay ahead of 5 us kernels).  Eight such threads started without affinity land wherever the scheduler puts them --
possibly all on one socket, across the inter-socket link from half of the GPUs' doorbells.  The reference's only equivalent is accelerate's
per-process launch (train.py:58-64), which does not pin either.

Pure host logic over sysfs, no GPU / HIP call (must run BEFORE the process touches the GPU; never execs):

  GPU ordinal -> KFD topology node (``/sys/class/kfd/kfd/topology/nodes/<n>/properties``: the nodes with ``simd_count > 0`` in node order are the
  HIP ordinals 0..N-1, after ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES filtering) -> ``drm_render_minor`` ->
  ``/sys/class/drm/renderD<minor>/device/numa_node`` -> ``/sys/devices/system/node/node<k>/cpulist``.

  The ranks that share a NUMA node split its CPUs (intersected with the process's current affinity mask, so a container's cpuset is respected)
  into disjoint contiguous slices, one per rank, in local-rank order.

Anything missing (no KFD topology, numa_node = -1, a single-node box, an empty slice) degrades to "leave the affinity alone" with the reason recorded.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]."""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def format_cpulist(cpus: Sequence[int]) -> str:
    cpus = sorted(set(cpus))
    if not cpus:
        return ""
    runs, a, prev = [], cpus[0], cpus[0]
    for c in cpus[1:]:
        if c != prev + 1:
            runs.append((a, prev)); a = c
        prev = c
    runs.append((a, prev))
    return ",".join(str(x) if x == y else f"{x}-{y}" for x, y in runs)


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _visible_filter(n_gpus: int, env: Dict[str, str]) -> List[int]:
    """Physical GPU indices behind HIP ordinals 0..: ROCR_VISIBLE_DEVICES filters first, then HIP_ / CUDA_VISIBLE_DEVICES index into what is left.
    Only plain integer lists are understood (UUID forms fall back to the identity)."""
    idx = list(range(n_gpus))
    for key in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(key)
        if v is None or key == "CUDA_VISIBLE_DEVICES" and "HIP_VISIBLE_DEVICES" in env:
            continue
        try:
            sel = [int(t) for t in v.split(",") if t.strip() != ""]
        except ValueError:
            continue
        idx = [idx[i] for i in sel if 0 <= i < len(idx)]
    return idx


def gpu_numa_nodes(sysfs: str = "/sys", env: Optional[Dict[str, str]] = None) -> Optional[List[int]]:
    """NUMA node of every visible GPU in HIP ordinal order (-1: unknown), or None when the KFD topology is not there."""
    env = dict(os.environ) if env is None else env
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted((int(d) for d in os.listdir(base) if d.isdigit()))
    except OSError:
        return None
    gpus = []
    for n in nodes:
        props = _read(os.path.join(base, str(n), "properties"))
        if props is None:
            continue
        kv = {}
        for line in props.splitlines():
            p = line.split()
            if len(p) == 2:
                kv[p[0]] = p[1]
        try:
            if int(kv.get("simd_count", "0")) <= 0:
                continue                                  # a CPU node
            minor = int(kv.get("drm_render_minor", "-1"))
        except ValueError:
            continue
        numa = -1
        if minor >= 0:
            t = _read(os.path.join(sysfs, "class", "drm", f"renderD{minor}", "device", "numa_node"))
            try:
                numa = int(t.strip()) if t is not None else -1
            except ValueError:
                numa = -1
        gpus.append(numa)
    if not gpus:
        return None
    return [gpus[i] for i in _visible_filter(len(gpus), env)]


def plan(local_rank: int, local_world: int, sysfs: str = "/sys", allowed: Optional[Sequence[int]] = None,
         env: Optional[Dict[str, str]] = None) -> dict:
    """The CPU set for local rank ``local_rank`` of ``local_world`` (rank r drives HIP ordinal r).  Returns
    {"cpus": [...] or None, "numa_node": k or None, "reason": str}; cpus None = do not touch the affinity."""
    if allowed is None:
        try:
            allowed = sorted(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            allowed = list(range(os.cpu_count() or 1))
    allowed = sorted(set(allowed))
    numa = gpu_numa_nodes(sysfs, env)
    if numa is None:
        return {"cpus": None, "numa_node": None, "reason": "no KFD topology under %s: affinity left alone" % sysfs}
    if local_rank >= len(numa):
        return {"cpus": None, "numa_node": None, "reason": "local rank %d but %d visible GPU(s): affinity left alone" % (local_rank, len(numa))}
    node = numa[local_rank]
    if node < 0:
        return {"cpus": None, "numa_node": None, "reason": "GPU %d reports numa_node -1 (single-node box or no ACPI affinity): affinity left alone" % local_rank}
    t = _read(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist"))
    if t is None:
        return {"cpus": None, "numa_node": node, "reason": "node%d has no cpulist: affinity left alone" % node}
    node_cpus = [c for c in parse_cpulist(t) if c in set(allowed)]
    sharers = [r for r in range(min(local_world, len(numa))) if numa[r] == node]           # local ranks on this node, in rank order
    k, n = sharers.index(local_rank), len(sharers)
    per = len(node_cpus) // n
    if per < 1:
        return {"cpus": None, "numa_node": node, "reason": "node%d offers %d allowed CPU(s) for %d rank(s): affinity left alone" % (node, len(node_cpus), n)}
    mine = node_cpus[k * per:(k + 1) * per] if k < n - 1 else node_cpus[k * per:]
    return {"cpus": mine, "numa_node": node, "reason": "GPU %d on NUMA node %d, slice %d of %d of its %d allowed CPUs" % (local_rank, node, k + 1, n, len(node_cpus))}


def apply(local_rank: int, local_world: int, sysfs: str = "/sys") -> dict:
    """plan() + sched_setaffinity on the calling process (all its future threads inherit it).  Call BEFORE any GPU / HIP call; never execs.
    PG_NO_AFFINITY=1 disables it.  Returns the plan with "applied" and the resulting "cpulist"."""
    if os.environ.get("PG_NO_AFFINITY") == "1":
        return {"cpus": None, "numa_node": None, "reason": "PG_NO_AFFINITY=1", "applied": False, "cpulist": None}
    p = plan(local_rank, local_world, sysfs)
    p["applied"] = False
    if p["cpus"]:
        try:
            os.sched_setaffinity(0, p["cpus"])
            p["applied"] = True
        except (AttributeError, OSError) as ex:
            p["reason"] += "; sched_setaffinity failed: %r" % (ex,)
    try:
        p["cpulist"] = format_cpulist(sorted(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        p["cpulist"] = None
    p.pop("cpus", None)
    return p
