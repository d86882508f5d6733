"""CPU: the host logic of bench.py that only matters with more than one rank - and has therefore never run on this pool's one-GPU boxes:
`python bench.py --gpus N` starting its own ranks, and the agreement of the ranks on a captured sharded step (a rank whose
capture failed must not wait in an all-reduce while the others replay a graph full of all-gathers)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_gpus_n_without_a_launcher_prints_the_child_command():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    cmd = d["launch"]
    assert d["ranks"] == 2
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]   # the ranks get the caller's arguments, minus --dry-launch


def test_dry_launch_under_a_launcher_starts_nothing():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert json.loads(out.stdout.strip().splitlines()[-1])["launch"] is None


def test_peer_store_child_group_command_and_its_clean_environment(monkeypatch):
    """The separate group of ranks that measures the peer-store all-gather on an N > 1 line: a torch.distributed.run child of its own, on a port
    of its own, carrying none of the launcher variables of the run that starts it (it would otherwise join the parent's rendezvous)."""
    sys.path.insert(0, ROOT)
    import bench
    d = bench.peer_store_child(8, 20, 5, dry=True)
    cmd = d["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--peer-child", "--gpus", "8", "--steps", "10", "--warmup", "2"]
    seen = {}

    class FakeProc:
        pid, returncode, hang = 4242, 0, False
        stdout_text = 'noise\n{"peer_allgather": {"ranks": 8, "peer_store": {"value": 1.0}}}\n'

        def __init__(self, cmd, env, **kw):
            seen.update(env=env, kw=kw)

        def communicate(self, timeout=None):
            if FakeProc.hang and timeout is not None:
                raise sp.TimeoutExpired("x", timeout)
            return FakeProc.stdout_text, ""
    import subprocess as sp
    monkeypatch.setattr(sp, "Popen", FakeProc)
    killed = []
    monkeypatch.setattr(os, "killpg", lambda pgid, sig: killed.append(pgid))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        monkeypatch.setenv(k, "7")
    assert bench.peer_store_child(8, 20, 5) == {"ranks": 8, "peer_store": {"value": 1.0}}
    assert not any(k in seen["env"] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID"))
    assert seen["env"]["FP8MI_BENCH_CHILD"] == "1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["kw"]["start_new_session"] is True     # a process group of its own: a timeout kills exactly that group
    assert bench.PEER_CHILD_TIMEOUT_S <= 480            # shorter than the 10 minutes the other ranks' rendezvous waits for rank 0
    FakeProc.stdout_text = "no line here\n"
    assert "error" in bench.peer_store_child(8, 20, 5) and not killed
    FakeProc.hang = True
    assert "killed" in bench.peer_store_child(8, 20, 5)["error"] and killed == [4242]


def test_headline_gather_is_chosen_on_evidence_only():
    sys.path.insert(0, ROOT)
    import bench
    good = {"backend": "rccl", "collective": {"value": 900.0, "unit": "TFLOP/s"},
            "peer_store": {"value": 2400.0, "bit_equal_to_collective_on_every_rank": True, "timeout_status": 0}}
    use, why = bench.choose_gather(good)
    assert use and "2400.0" in why
    for breakit in (lambda d: d["peer_store"].update(value=930.0),                                    # not clearly faster
                    lambda d: d["peer_store"].update(bit_equal_to_collective_on_every_rank=False),
                    lambda d: d["peer_store"].update(timeout_status=2),
                    lambda d: d.update(backend="gloo"),                                               # a rehearsal is not a measurement
                    lambda d: d.update(peer_store={"error": "boom"}),
                    lambda d: d.pop("collective")):
        d = json.loads(json.dumps(good))
        breakit(d)
        use, why = bench.choose_gather(d)
        assert not use and why
    assert bench.choose_gather({"error": "the child group printed no line"})[0] is False


def test_nccl_debug_log_summary():
    import bench
    log = "\n".join([
        "host:1:1 [0] NCCL INFO RCCL version 2.22.3+hip6.4",
        "host:1:1 [0] NCCL INFO comm 0xabc rank 0 nranks 8 cudaDev 0 busId c000 - Init START",
        "host:1:1 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC",
        "host:1:1 [0] NCCL INFO Channel 01/0 : 0[0] -> 2[2] via P2P/IPC",
        "host:1:1 [0] NCCL INFO Channel 00/0 : 0[0] -> 9[1] [send] via NET/Socket/0",
        "host:1:1 [0] NCCL INFO Connected all rings",
        "host:1:1 [0] NCCL INFO 16 coll channels, 16 collnet channels, 0 nvls channels, 16 p2p channels",
    ])
    d = bench.parse_nccl_debug(log)
    assert d["transports"]["P2P/IPC"] == 2 and d["transports"]["NET/Socket/0"] == 1
    assert any("RCCL version" in ln for ln in d["selected_lines"]) and any("coll channels" in ln for ln in d["selected_lines"])


def _agree_worker(rank, world, port, case, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        replays = []

        def replay(g):
            replays.append(g)
            if case == "replay_fails_on_rank1":
                # (an error BEFORE any captured collective ran - e.g. graph instantiation; once a collective is under way a lost rank is a job failure)
                if rank == 1:
                    raise RuntimeError("injected replay failure")
                return
            # a captured step holds collectives: every rank that replays must find its peers replaying too
            t = torch.ones(1)
            dist.all_reduce(t)
            assert int(t.item()) == world
        ok = not (case == "capture_fails_on_rank1" and rank == 1)
        graph = object() if ok else None
        got = bench.agree_on_graph(ok, graph, replay, "cpu")
        # whatever was agreed, the ranks are in step afterwards: one more collective completes
        t = torch.ones(1)
        dist.all_reduce(t)
        q.put((rank, got is not None, len(replays), int(t.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,expect_graph,expect_replays", [("all_ok", True, 1), ("capture_fails_on_rank1", False, 0), ("replay_fails_on_rank1", False, None)])
def test_ranks_agree_before_a_captured_sharded_step_is_replayed(case, expect_graph, expect_replays):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, has_graph, n_replays, s in res:
        assert has_graph == expect_graph, (case, res)     # the SAME verdict on every rank
        assert s == world                                 # and nobody is stuck in a different collective
        if expect_replays is not None:
            assert n_replays == expect_replays, (case, res)   # a rank whose peer failed to capture never replays
