"""CPU-only, world size 2-3: the host logic of pp_ps_migrate / pp_allreduce_sum that needs no GPU --
the id broadcast (pp_bootstrap_broadcast), the count exchange (PS_Comm_Ialltoall, SCS_migrate.h:48),
the offsets both sides derive from the counts (pp_migrate_plan, SCS_migrate.h:66-72,129-133) and the
host collectives of the built-in TCP transport and of a caller-supplied transport (gloo through
ctypes callbacks, the way an MPI build would plug MPI in).  The device side of the same calls is
covered by the -m gpu tests (local communicator with virtual ranks, two processes on one GPU)."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _expected_counts(world):
    """send[r][q]: particles rank r sends to rank q (diagonal ignored by the protocol)"""
    rng = np.random.default_rng(11)
    m = rng.integers(0, 50, size=(world, world)).astype(np.int32)
    m[0, world - 1] = 0  # an empty pair
    return m


def _tcp_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        import pumipic_amd_loader
        pumipic_amd_loader.load()
        from pumipic_amd import capi
        L = capi.lib()
        # 1. the id broadcast
        import ctypes as C
        payload = (C.c_ubyte * 128)()
        if rank == 0:
            for i in range(128):
                payload[i] = (7 * i + 3) % 251
        capi.check(L.pp_bootstrap_broadcast(b"127.0.0.1", port, rank, world, payload, 128))
        assert bytes(payload) == bytes((7 * i + 3) % 251 for i in range(128))
        # 2. a TCP communicator: counts, plan, host all-reduce, barrier
        comm = capi.Comm.tcp("127.0.0.1", port + 1, rank, world)
        assert comm.kind() == "tcp" and comm.rank() == rank and comm.size() == world
        m = _expected_counts(world)
        recv = comm.exchange_counts(m[rank])
        want = m[:, rank].copy()
        assert np.array_equal(recv, want), (recv, want)
        sd, rd, ns, nr = capi.migrate_plan(world, rank, m[rank], recv)
        others = [r for r in range(world) if r != rank]
        assert ns == int(m[rank, others].sum()) and nr == int(m[others, rank].sum())
        run_s = run_r = 0
        for r in range(world):
            assert sd[r] == run_s and rd[r] == run_r
            if r != rank:
                run_s += m[rank, r]
                run_r += m[r, rank]
        tot = comm.allreduce_sum_host([rank + 1, 10 * (rank + 1)])
        assert list(tot) == [sum(range(1, world + 1)), 10 * sum(range(1, world + 1))]
        comm.barrier()
        comm.destroy()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + repr(e) + traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_tcp_transport_host_logic(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tcp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


def _gloo_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        import pumipic_amd_loader
        pumipic_amd_loader.load()
        from pumipic_amd import capi
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = capi.Comm.torch()
        assert comm.kind() == "host"
        m = _expected_counts(world)
        recv = comm.exchange_counts(m[rank])
        assert np.array_equal(recv, m[:, rank])
        tot = comm.allreduce_sum_host([rank + 5])
        assert int(tot[0]) == sum(r + 5 for r in range(world))
        comm.barrier()
        comm.destroy()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + repr(e) + traceback.format_exc()))


def test_host_transport_over_gloo_callbacks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res


def test_migrate_plan_rejects_bad_input(pp):
    from pumipic_amd import capi
    capi.build()
    with pytest.raises(capi.PPError):
        capi.migrate_plan(2, 0, [0, -1], [0, 0])
    sd, rd, ns, nr = capi.migrate_plan(1, 0, [9], [9])
    assert ns == 0 and nr == 0  # a rank keeps its own particles


def test_single_rank_env_communicator(pp, monkeypatch):
    from pumipic_amd import capi
    capi.build()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    c = capi.Comm.env()
    assert c.kind() == "self" and c.size() == 1 and c.rank() == 0
    assert list(c.exchange_counts([4])) == [4]
    c.destroy()


def test_missing_librccl_is_an_error_not_a_crash():
    """A box without (a usable) librccl: pp_comm_unique_id / pp_comm_create_rccl must fail with a
    message (the library's 'works without RCCL' contract, bench.py's fallback to --comm torch relies on
    it) -- not die inside strlen(NULL) because dlerror() was consumed twice.  Runs in a child process:
    the loader remembers its verdict for the life of the process."""
    import subprocess
    code = r"""
import os, sys
sys.path.insert(0, %r)
os.environ["PP_RCCL_LIB"] = "/nonexistent/librccl_not_here.so"
import pumipic_amd_loader
pumipic_amd_loader.load()
from pumipic_amd import capi
for attempt in range(2):   # the second call takes the 'already tried' branch
    try:
        capi.Comm.unique_id()
    except capi.PPError as e:
        assert "librccl could not be opened" in str(e) and "librccl_not_here" in str(e), str(e)
    else:
        raise SystemExit("pp_comm_unique_id succeeded without a library")
print("ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout, r.stderr[-2000:])


def test_tcp_rank0_gives_up_on_a_peer_that_never_comes():
    """rank 0 of the built-in TCP transport accepts with the same deadline its peers connect with:
    a rank that died before the rendezvous is an error after PP_COMM_TIMEOUT seconds, not a hang."""
    import subprocess
    import time
    code = r"""
import os, sys
sys.path.insert(0, %r)
os.environ["PP_COMM_TIMEOUT"] = "2"
import pumipic_amd_loader
pumipic_amd_loader.load()
from pumipic_amd import capi
try:
    capi.Comm.tcp("127.0.0.1", %d, 0, 2)
except capi.PPError as e:
    assert "waited" in str(e) and "peers" in str(e), str(e)
    print("ok")
else:
    raise SystemExit("a 2-rank communicator formed with one rank")
""" % (ROOT, _free_port())
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout, r.stderr[-2000:])
    assert time.time() - t0 < 60


def test_bench_watchdog_exits_with_code_3():
    """bench.py's Watchdog (armed on multi-rank runs): no beat for `limit` seconds -> the process exits with
    code 3 and says which phase it was in; beats keep it alive."""
    import subprocess
    code = r"""
import sys, time
sys.path.insert(0, %r)
import bench
bench.WATCHDOG = bench.Watchdog(1.0, 7)
for i in range(8):          # 2.4 s of steady progress: still alive
    time.sleep(0.3)
    bench.beat("step %%d" %% i)
print("alive", flush=True)
bench.beat("the exchange")
time.sleep(30)              # ... then silence
print("not reached")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr[-1500:])
    assert "alive" in r.stdout and "not reached" not in r.stdout
    assert "rank 7 made no progress" in r.stderr and "the exchange" in r.stderr


def test_mirror_distributor_rank_subset_host_logic(pp, tmp_path):
    """support/psDistributor.hpp:10-138 in the mirror headers (pumi-pic_amd/include/particle_structs.hpp): the
    world form and the rank-subset form, host side -- tests/cpp/distributor_host.cpp, compiled with hipcc and
    run without a GPU (no HIP call is made).  What a subset means for a migration is a GPU test
    (tests/test_gpu_comm.py::test_cpp_driver_distributor_rank_subset_is_enforced)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    libdir = os.path.join(ROOT, "pumi-pic_amd")
    exe = str(tmp_path / "distributor_host")
    # (host code of the mirror headers under AddressSanitizer on this CPU build)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fsanitize=address",
                           "-fno-omit-frame-pointer",
                           os.path.join(ROOT, "tests", "cpp", "distributor_host.cpp"), "-o", exe,
                           "-L" + libdir, "-lpumipic_hip", "-Wl,-rpath," + libdir])
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    env["ASAN_OPTIONS"] = "detect_leaks=0"
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=env)
    assert "AddressSanitizer" not in out.stdout, out.stdout[-3000:]
    assert out.returncode == 0 and "all checks passed" in out.stdout, out.stdout


def test_mpi_facade_two_processes_host_only(pp, tmp_path):
    """pumi-pic_amd/include/pumipic_mpi.hpp: MPI_Comm_rank / size, MPI_Allreduce (every type, SUM / MAX / MIN, several
    values), MPI_Reduce (root only), MPI_Barrier over the library's communicator -- two processes on this machine, the
    host-staged transport, no GPU (tests/cpp/mpi_facade_host.cpp)."""
    import shutil
    import socket
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    libdir = os.path.join(ROOT, "pumi-pic_amd")
    exe = str(tmp_path / "mpi_facade_host")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fsanitize=address",
                           "-fno-omit-frame-pointer",
                           os.path.join(ROOT, "tests", "cpp", "mpi_facade_host.cpp"), "-o", exe,
                           "-L" + libdir, "-lpumipic_hip", "-Wl,-rpath," + libdir])
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PP_COMM="tcp", PP_COMM_PORT=str(port), ASAN_OPTIONS="detect_leaks=0")
        env.pop("LD_PRELOAD", None)
        procs.append(subprocess.Popen([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert "AddressSanitizer" not in o, o[-3000:]
        assert p.returncode == 0 and ("rank %d: all checks passed" % r) in o, o
        assert ("rank %d of 2" % r) in o
