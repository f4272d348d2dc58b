"""CPU-side tests (-m "not gpu"): the C-ABI library loads and exports every symbol that
include/pbbi.h declares (no compute calls), the host mirror of the reference interface
behaves like the reference (known answers held by the reference's own asserting tests),
and the N > 1 sharding path runs under gloo with world_size 2."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pbbi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(pbbi_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    from physicsbasedbayesianinference_amd import _lib
    lib = _lib.load()
    assert sorted(_lib.PROTOTYPES) == declared, "ctypes table and header drifted apart"
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pbbi_version() == 103
    # the symbols are really exported by the shared object (not resolved from elsewhere)
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True,
                         text=True).stdout
    exported = set(re.findall(r"\bT (pbbi_[a-z0-9_]+)", out))
    assert set(declared) <= exported


def test_library_holds_gfx950_code_only():
    from physicsbasedbayesianinference_amd import _lib
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          "--input=" + _lib.LIB_PATH], capture_output=True, text=True).stdout
    if out.strip():  # bundler understands the fat binary section on this toolchain
        targets = [l for l in out.split() if "amdgcn" in l]
        assert targets and all("gfx950" in t for t in targets), targets


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import physicsbasedbayesianinference_amd as P
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.StandardGaussian(2)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "physicsbasedbayesianinference_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "liboracle" not in txt, f


def test_ensemble_known_answers():
    """src/tests/test_ensemble.py:26-44 and the IndexError check of its main()."""
    from physicsbasedbayesianinference_amd import Ensemble
    g = load_golden("G10_known_answers")
    e = Ensemble(4, 100)
    q1, p1, m1, w1 = e.particle(10)
    assert np.all(q1 == 0) and np.all(p1 == 0) and m1 == 1.0 and w1 == 0.0
    assert np.array_equal(q1, g["particle_q"]) and m1 == float(g["particle_m"])
    with pytest.raises(IndexError) as ei:
        e.particle(101)
    assert str(ei.value) == str(g["index_error"])
    assert e.q.shape == (4, 100) and e.q.flags.c_contiguous and e.q.dtype == np.float64


def test_ensemble_rng_matches_reference_stream():
    """setPosition / setMomentum consume the global RandomState exactly like the reference
    (compare with q0 / p_draw recorded from the reference run)."""
    from physicsbasedbayesianinference_amd import Ensemble
    for name in ("G3_getsamples_c1", "G8_getsamples_mass", "G11_getsamples_test2"):
        g = load_golden(name)
        np.random.seed(int(g["seed"]))
        e = Ensemble(int(g["D"]), int(g["N"]))
        e.mass = g["mass"].copy()
        q_before = e.q
        q = e.setPosition(float(g["qStd"]))
        assert q is e.q and q is not q_before          # rebinding
        assert np.array_equal(q, g["q0"])
        p = e.setMomentum(float(g["temperature"]))
        assert p is e.p and np.array_equal(p, g["p_draw"][0])
        u = np.random.uniform(size=int(g["N"]))
        assert np.array_equal(u, g["u"][0])


class _FakePot:
    """Host-logic stand-in so that constructor paths can be exercised without a GPU."""


def _fake_potential(D):
    from physicsbasedbayesianinference_amd.potential import Potential
    pot = Potential.__new__(Potential)
    pot.numDimensions, pot.dtype, pot.device = D, np.dtype("float64"), 0
    import ctypes
    pot._handle = ctypes.c_void_p()
    return pot


def test_numsteps_and_constructor_logic():
    from physicsbasedbayesianinference_amd import HMC, Ensemble, Integrator, Leapfrog, StormerVerlet
    g = load_golden("G9_numsteps")
    pot = _fake_potential(1)
    ens = Ensemble(1, 3)
    for T, h, n in zip(g["finalTime"], g["stepSize"], g["numSteps"]):
        assert Integrator(ens, float(h), float(T), pot.gradient).numSteps == int(n)
    integ = Leapfrog(ens, 0.1, 1.0, pot)
    assert integ.q is ens.q and integ.p is ens.p and integ.mass is ens.mass  # aliases
    assert integ.potential is pot
    with pytest.raises(NotImplementedError):
        Integrator(ens, 0.1, 1.0, pot.gradient).integrate()
    with pytest.raises(NotImplementedError):
        Leapfrog(ens, 0.1, 1.0, None)            # N-body mode is out of scope
    with pytest.raises(TypeError, match="no CPU fallback"):
        Leapfrog(ens, 0.1, 1.0, lambda q: np.tanh(q))   # callables that cannot be traced are rejected
    with pytest.raises(ValueError, match="Invalid integration method selected."):
        HMC(ens, 1.0, 0.1, None, potential=pot, method="RK4")
    h = HMC(ens, 0.5, 0.05, pot.density, method="Stormer-Verlet")
    assert isinstance(h.integrator, StormerVerlet) and h.integrator.numSteps == 10
    assert h.potential == h.potentialFunc


def test_shard_bounds_cover_the_ensemble():
    from physicsbasedbayesianinference_amd.distributed import shard_bounds
    for N in (0, 1, 7, 64, 65536, 524288 + 3):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(N, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == N
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from oracle import oracle as orc
from physicsbasedbayesianinference_amd.distributed import HostStream, gather_samples, shard_bounds
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
rank, world = dist.get_rank(), dist.get_world_size()
D, N, S, L, h, seed = 5, {N}, 3, 10, 0.1, 99
rs = np.random.RandomState(0); A = rs.standard_normal((D, D)); Pm = np.linalg.inv(A @ A.T / D + np.eye(D)); Pm = 0.5 * (Pm + Pm.T)
pot = orc.pot_gauss_dense(np.zeros(D), Pm)
lo, hi = shard_bounds(N, rank, world)
# --- numpy-stream mode: every rank replays the global stream and keeps its columns
np.random.seed(seed)
hs = HostStream(D, N, lo, hi)
q = hs.positions(1.0)
slabs = []
for i in range(S):
    p = hs.momenta(np.ones(hi - lo), 1.0 / 1.380649e-23); u = hs.uniforms()
    orc.hmc_iter(pot, "Leapfrog", q, p, u, None, h, L)   # the oracle stands in for the GPU kernels on CPU
    slabs.append(q.copy())
full = gather_samples(torch.from_numpy(np.stack(slabs))).numpy()
ref = orc.get_samples_numpy_stream(pot, "Leapfrog", D, N, S, 1.0, h, 1.0 / 1.380649e-23, 1.0, seed)
assert full.shape == (S, D, N)
assert np.array_equal(np.transpose(full, (1, 2, 0)), ref["samples"]), "numpy-stream sharding differs"
# --- philox mode: global chain index in the counter
q = orc.philox_normal(seed, orc.STREAM_POSITION, 0, lo, D, hi - lo)
s_loc, _, rej, _ = orc.hmc_run_philox(pot, "Leapfrog", q, None, h, L, S, seed=seed, chain0=lo)
full = gather_samples(torch.from_numpy(s_loc)).numpy()
q = orc.philox_normal(seed, orc.STREAM_POSITION, 0, 0, D, N)
s_ref, _, _, _ = orc.hmc_run_philox(pot, "Leapfrog", q, None, h, L, S, seed=seed, chain0=0)
assert np.array_equal(full, s_ref), "philox sharding differs"
# --- ensemble weights: all-reduce(MIN) + all-reduce(SUM) == the single-process normalisation
from physicsbasedbayesianinference_amd.distributed import ensemble_weights
H = np.random.RandomState(5).standard_normal(N) * 30.0 + 800.0          # exp(-H) underflows unshifted
w_loc, logZ = ensemble_weights(torch.from_numpy(H[lo:hi].copy()))
w_ref = np.exp(-(H - H.min())); Z = w_ref.sum(); w_ref /= Z
assert np.allclose(w_loc.numpy(), w_ref[lo:hi], rtol=1e-13) and abs(logZ - (np.log(Z) - H.min())) < 1e-10
# --- chunked, OVERLAPPED collection (distributed.OverlappedGather: chunk k's all-gather in flight while chunk
#     k+1 "samples"; two send / two receive buffers, received in place) == the single-process run, both RNG modes
from physicsbasedbayesianinference_amd.distributed import OverlappedGather, gather_blocks
S2, chunk = 7, 3     # chunks of 3, 3, 1 iterations
ref2 = orc.get_samples_numpy_stream(pot, "Leapfrog", D, N, S2, 1.0, h, 1.0 / 1.380649e-23, 1.0, seed)
q0 = orc.philox_normal(seed, orc.STREAM_POSITION, 0, 0, D, N)
s_ref2, m_ref2, _, _ = orc.hmc_run_philox(pot, "Leapfrog", q0, None, h, L, S2, seed=seed, chain0=0)
for mode in ("numpy", "philox"):
    og = OverlappedGather((chunk, D, hi - lo), torch.float64, "cpu", N)
    got = []
    def consume(blk):
        assert blk.blocks.shape[0] == world and blk.sizes == [shard_bounds(N, r, world)[1] - shard_bounds(N, r, world)[0] for r in range(world)]
        full = blk.to_sdn().numpy().copy()
        v4 = blk.view4()                                   # (c, D, world, Nmax) view of the receive buffer itself
        assert v4.data_ptr() == blk.blocks.data_ptr()
        for r in range(world):
            assert np.array_equal(v4[:, :, r, :blk.sizes[r]].numpy(), full[:, :, blk.offsets[r]:blk.offsets[r] + blk.sizes[r]])
        got.append(full)
    if mode == "numpy":
        np.random.seed(seed); hs = HostStream(D, N, lo, hi); q = hs.positions(1.0)
    else:
        q = orc.philox_normal(seed, orc.STREAM_POSITION, 0, lo, D, hi - lo)
    done = 0
    for k in range((S2 + chunk - 1) // chunk):
        c = min(chunk, S2 - done)
        buf = og.local(k)
        for i in range(c):
            if mode == "numpy":
                p = hs.momenta(np.ones(hi - lo), 1.0 / 1.380649e-23); u = hs.uniforms()
                orc.hmc_iter(pot, "Leapfrog", q, p, u, None, h, L)
            else:
                orc.hmc_run_philox(pot, "Leapfrog", q, None, h, L, 1, seed=seed, iter0=done + i, chain0=lo)
            buf[i] = torch.from_numpy(q)
        blk = og.submit(k, c)
        if blk is not None:
            consume(blk)
        done += c
    consume(og.finish())
    full = np.concatenate(got)
    want = np.transpose(ref2["samples"], (2, 0, 1)) if mode == "numpy" else s_ref2
    assert full.shape == (S2, D, N) and np.array_equal(full, want), "overlapped chunks differ: " + mode
# --- gather_blocks with n_total: no size exchange, received in place; gather_samples re-lays in bounded pieces
loc = torch.from_numpy(s_loc)
blk = gather_blocks(loc, n_total=N)
assert np.array_equal(blk.to_sdn().numpy(), s_ref) and blk.blocks.shape == (world, S, D, max(blk.sizes))
assert np.array_equal(gather_samples(loc, n_total=N, max_chunk_bytes=1).numpy(), s_ref)   # one slab per collective
assert np.array_equal(gather_samples(loc).numpy(), s_ref)                                 # sizes asked from the ranks
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("N", [64, 37])
def test_two_rank_gloo_sharding_matches_single_process(tmp_path, N):
    """world_size = 2 on CPU (gloo): shard -> iterate -> ONE all-gather reproduces the
    single-process ensemble bit for bit, for both RNG modes and for an uneven split."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, port=port, N=N))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


# ------------------------------------------------------------------ user-defined potentials
def test_custom_potential_plugin_builds_and_exports():
    """custom.compile_plugin: the generated translation unit compiles for gfx950 without a GPU and
    the plugin exports the entry points pbbi_potential_create_custom resolves (this also leaves
    the plugins of the GPU parity tests in the in-tree cache, which travels to the GPU box)."""
    import subprocess
    from physicsbasedbayesianinference_amd.custom import compile_plugin
    from custom_sources import COIN_TOSS_AD, COIN_TOSS_SOURCE, LOGISTIC, LOGISTIC_AD, QUARTIC, QUARTIC_AD
    # (source, dtype, D): the dimensions the GPU parity tests construct (D <= 16 in fp64 / 32 in fp32
    # compile the register-resident kernels as well), and one build without a fixed dimension
    for src, dtype, D in ((QUARTIC, "float64", 11), (QUARTIC, "float64", 40), (LOGISTIC, "float64", 5),
                          (QUARTIC, "float32", 7), (COIN_TOSS_SOURCE, "float64", 1),
                          (QUARTIC, "float64", 9), (QUARTIC, "float64", 20), (QUARTIC, "float64", 24),
                          (QUARTIC, "float64", 12), (COIN_TOSS_SOURCE, "float64", 2), (QUARTIC, "float64", 48),
                          # sources without a gradient: dual-number autodiff (csrc/pbbi_autodiff.h)
                          (QUARTIC_AD, "float64", 11), (QUARTIC_AD, "float64", 40), (LOGISTIC_AD, "float64", 5),
                          (COIN_TOSS_AD, "float64", 2), (QUARTIC_AD, "float32", 7),
                          (QUARTIC, "float64", None)):
        so = compile_plugin(src, dtype, D=D)
        syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
        for name in ("pbbi_plugin_abi", "pbbi_plugin_dtype", "pbbi_plugin_hmc_iter",
                     "pbbi_plugin_integrate", "pbbi_plugin_eval"):
            assert f" T {name}" in syms, name
        assert compile_plugin(src, dtype, D=D) == so  # cache hit
    with pytest.raises(RuntimeError, match="hipcc failed"):
        compile_plugin("this is not C++")


def test_autodiff_source_on_the_host():
    """The dual-number text (csrc/pbbi_autodiff.h) compiled for the HOST by the oracle's g++ build:
    the automatic gradient of the gradient-less sources equals the hand-written ones."""
    from oracle import oracle as orc
    from physicsbasedbayesianinference_amd.custom import complete_source, has_gradient
    from custom_sources import LOGISTIC, LOGISTIC_AD, QUARTIC, QUARTIC_AD, logistic_problem
    assert has_gradient(QUARTIC) and not has_gradient(QUARTIC_AD)
    rs = np.random.RandomState(1)
    q = rs.standard_normal((6, 9))
    U0, g0 = orc.potential(orc.pot_custom(complete_source(QUARTIC), 6, [2.0, 0.5]), q, want_grad=True)
    U1, g1 = orc.potential(orc.pot_custom(complete_source(QUARTIC_AD), 6, [2.0, 0.5]), q, want_grad=True)
    assert np.array_equal(U0, U1) and np.allclose(g0, g1, rtol=1e-13, atol=1e-14)
    X, y, lam, prm = logistic_problem()
    q = rs.standard_normal((X.shape[1], 7))
    U0, g0 = orc.potential(orc.pot_custom(complete_source(LOGISTIC), X.shape[1], prm), q, want_grad=True)
    U1, g1 = orc.potential(orc.pot_custom(complete_source(LOGISTIC_AD), X.shape[1], prm), q, want_grad=True)
    assert np.allclose(U0, U1, rtol=1e-13) and np.allclose(g0, g1, rtol=1e-11, atol=1e-12)


def test_oracle_custom_potential_matches_numpy():
    """oracle.pot_custom (the same user source compiled for the host) against closed forms."""
    from oracle import oracle as orc
    from custom_sources import LOGISTIC, QUARTIC, logistic_numpy, logistic_problem
    rs = np.random.RandomState(0)
    q = rs.standard_normal((6, 9))
    U, g = orc.potential(orc.pot_custom(QUARTIC, 6, [2.0, 0.5]), q, want_grad=True)
    d = q[:-1] - q[1:]
    assert np.allclose(U, 0.5 * (q ** 4).sum(0) + 0.25 * (d * d).sum(0), rtol=1e-13)
    gc = 2.0 * q ** 3
    gc[:-1] += 0.5 * d
    gc[1:] -= 0.5 * d
    assert np.allclose(g, gc, rtol=1e-13, atol=1e-14)
    X, y, lam, prm = logistic_problem()
    q = rs.standard_normal((X.shape[1], 7))
    U, g = orc.potential(orc.pot_custom(LOGISTIC, X.shape[1], prm), q, want_grad=True)
    Un, gn = logistic_numpy(X, y, lam, q)
    assert np.allclose(U, Un, rtol=1e-12) and np.allclose(g, gn, rtol=1e-11, atol=1e-12)


# ------------------------------------------------------------------ bench.py launcher logic (no GPU)
def test_bench_self_launches_the_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts N ranks itself (a child torchrun on
    127.0.0.1, created before anything touches the GPU) and passes the child's exit code through; with
    a launcher of another world size it refuses instead of printing single-GPU numbers for every N."""
    import importlib
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    calls = []

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        calls.append((cmd, env))
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert bench.main() == 7
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a launcher that started another number of ranks: error, not a silent single-GPU run
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    assert bench.main() == 2
    assert len(calls) == 1


def test_warmup_key_and_flag_values_match_the_header():
    """The Python constants mirror include/pbbi.h (flags, streams) and the warm-up key differs from any seed."""
    from physicsbasedbayesianinference_amd import _lib
    from physicsbasedbayesianinference_amd.HMC import WARMUP_SEED_MASK
    hdr = open(os.path.join(ROOT, "include", "pbbi.h")).read()
    for name, val in (("PBBI_COMPAT_P_FROM_OLDQ", _lib.COMPAT_P_FROM_OLDQ), ("PBBI_KDK_FMA", _lib.KDK_FMA),
                      ("PBBI_BETA_ACCEPT", _lib.BETA_ACCEPT), ("PBBI_PER_CHAIN_STEPS", _lib.PER_CHAIN_STEPS),
                      ("PBBI_UTURN_STOP", _lib.UTURN_STOP), ("PBBI_STREAM_STEPS", _lib.STREAM_STEPS),
                      ("PBBI_DRAW_F64", _lib.DRAW_F64)):
        assert re.search(r"\b%s\s*=\s*%d\b" % (name, val), hdr), name
    assert re.search(r"\bPBBI_STREAM_DRAW_F64\s*=\s*0x%x\b" % _lib.STREAM_DRAW_F64, hdr)
    assert WARMUP_SEED_MASK != 0 and (5 ^ WARMUP_SEED_MASK) != 5


# ------------------------------------------------------------------ the fast host stream (csrc/hoststream.c)
def test_fast_host_stream_is_numpys_legacy_stream_bit_for_bit(monkeypatch):
    """_hoststream.standard_normal / uniform return exactly np.random's legacy draws from the same
    global state and leave exactly NumPy's state behind (key, position, cached gaussian): odd and even
    counts, a cached variate at entry, block boundaries of MT19937, interleaving with np.random calls."""
    from physicsbasedbayesianinference_amd import _hoststream as hs
    assert hs.available(), "libpbbi_host.so missing: python -m physicsbasedbayesianinference_amd.build"
    monkeypatch.setattr(hs, "MIN_FAST", 1)

    def same_state(a, b):
        return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    for seed in (0, 1234, 2 ** 32 - 1):
        for sizes in ([1, 2, 3, 5, 8], [1000, 1001, 77], [624 * 3, 313, 1, 1, 99999], [(7, 4001)], [262144 + 1]):
            np.random.seed(seed)
            ref = []
            for i, n in enumerate(sizes):
                ref += [np.random.standard_normal(n), np.random.uniform(size=int(np.prod(n)) // 3 + i + 1)]
            st_ref = np.random.get_state()
            np.random.seed(seed)
            got = []
            for i, n in enumerate(sizes):
                got += [hs.standard_normal(n), hs.uniform(int(np.prod(n)) // 3 + i + 1)]
            assert all(np.array_equal(a, b) and a.shape == b.shape for a, b in zip(ref, got)), (seed, sizes)
            assert same_state(st_ref, np.random.get_state()), (seed, sizes)
    # a NumPy draw that leaves a cached gaussian, then the fast path, then NumPy again
    np.random.seed(3)
    ref = [np.random.standard_normal(3), np.random.standard_normal(100001), np.random.uniform(size=7),
           np.random.standard_normal(2)]
    np.random.seed(3)
    got = [np.random.standard_normal(3), hs.standard_normal(100001), np.random.uniform(size=7),
           np.random.standard_normal(2)]
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))


def test_fast_host_stream_jump_ahead_generators(monkeypatch):
    """MT19937 jump-ahead (hoststream.c: phi by Berlekamp-Massey, x^n mod phi, Horner on the state): a jumped state
    equals the sequentially generated one in every bit the recurrence reads, and a draw whose blocks come from
    several generator threads -- each started from a jump, ranges of 1 .. 8 blocks, with and without shortfall
    passes -- hands out NumPy's values and NumPy's state bit for bit."""
    import ctypes as C
    from physicsbasedbayesianinference_amd import _hoststream as hs
    monkeypatch.setattr(hs, "MIN_FAST", 1)
    lib = hs._load()
    lib.pbbi_host_debug_jump_check.argtypes = [C.c_int64, C.c_uint32]
    lib.pbbi_host_debug_set_gen.argtypes = [C.c_int, C.c_int64, C.c_int64]
    lib.pbbi_host_debug_set_pass.argtypes = [C.c_double, C.c_int64, C.c_int]
    for blocks, seed in ((1, 1), (2, 77), (3, 5), (17, 123456789), (400, 4242)):
        assert lib.pbbi_host_debug_jump_check(blocks, seed) == 0, blocks
    threads0 = lib.pbbi_host_threads()
    try:
        for gens, min_blocks, rnd in ((4, 4, 2), (3, 2, 1), (8, 16, 8), (2, 8, 4)):
            lib.pbbi_host_debug_set_gen(gens, min_blocks, rnd)
            for threads, passes in ((8, (1.02, 1024, 15)), (16, (0.4, 3, 9)), (5, (1.02, 1024, 12))):
                lib.pbbi_host_set_threads(threads)
                lib.pbbi_host_debug_set_pass(*passes)
                for n in (1, 2, 1000, 4097, 65536, 300001, 1200000):
                    np.random.seed(n % 1000 + gens)
                    np.random.standard_normal(n % 13)          # odd positions, a cached variate
                    st0 = np.random.get_state()
                    ref, st_ref = np.random.standard_normal(n), np.random.get_state()
                    np.random.set_state(st0)
                    got, st = hs.standard_normal(n), np.random.get_state()
                    assert np.array_equal(got, ref), (gens, threads, n)
                    assert st[0] == st_ref[0] and np.array_equal(st[1], st_ref[1]) and st[2:] == st_ref[2:]
        # the shipped configuration on a draw large enough to use it (two generators from 8 192 new blocks on),
        # through the scaled form Ensemble.setMomentum uses
        lib.pbbi_host_debug_set_gen(2, 8192, 1024)
        lib.pbbi_host_debug_set_pass(1.02, 1024, 15)
        lib.pbbi_host_set_threads(8)
        D, N = 24, 250000
        scale = 1.0 + 0.25 * (np.arange(N) % 4)
        np.random.seed(99)
        ref, st_ref = np.random.standard_normal((D, N)) * scale, np.random.get_state()
        np.random.seed(99)
        out = np.empty((D, N))
        assert hs.scaled_normal_into(out, scale) is out
        st = np.random.get_state()
        assert np.array_equal(out, ref)
        assert st[0] == st_ref[0] and np.array_equal(st[1], st_ref[1]) and st[2:] == st_ref[2:]
    finally:
        lib.pbbi_host_debug_set_gen(2, 8192, 1024)
        lib.pbbi_host_debug_set_pass(1.02, 1024, 15)
        lib.pbbi_host_set_threads(threads0)


def test_fast_host_stream_shortfall_passes(monkeypatch):
    """normal_core's second and later passes (the first pass accepted fewer pairs than needed: acc0 / att0
    continuation, prefix array and word buffer regrown) are a > 100 sigma event with the shipped margin, so
    a test knob (pbbi_host_debug_set_pass) shrinks the request per pass and the chunk: values AND the
    MT19937 state handed back must still be NumPy's, whatever the number of passes and threads."""
    import ctypes as C
    from physicsbasedbayesianinference_amd import _hoststream as hs
    monkeypatch.setattr(hs, "MIN_FAST", 1)
    lib = hs._load()
    lib.pbbi_host_debug_set_pass.argtypes = [C.c_double, C.c_int64, C.c_int]
    lib.pbbi_host_debug_last_passes.restype = C.c_int
    threads0 = lib.pbbi_host_threads()
    try:
        seen = set()
        for factor, extra, log2ch in ((0.5, 0, 8), (0.3, 7, 6), (0.9, 0, 10), (0.05, 1, 4)):
            for threads in (1, 2, 5):
                lib.pbbi_host_set_threads(threads)
                lib.pbbi_host_debug_set_pass(factor, extra, log2ch)
                for seed, sizes, cached in ((5, [1000, 4097, 3], False), (6, [20001], True), (7, [(3, 777), 2], True)):
                    np.random.seed(seed)
                    if cached:
                        np.random.standard_normal(1)
                    ref = [np.random.standard_normal(n) for n in sizes] + [np.random.uniform(size=5)]
                    st_ref = np.random.get_state()
                    np.random.seed(seed)
                    if cached:
                        np.random.standard_normal(1)
                    got = []
                    for n in sizes:
                        got.append(hs.standard_normal(n))
                        seen.add(lib.pbbi_host_debug_last_passes())
                    got.append(hs.uniform(5))
                    assert all(np.array_equal(a, b) for a, b in zip(ref, got)), (factor, threads, seed)
                    st = np.random.get_state()
                    assert st[0] == st_ref[0] and np.array_equal(st[1], st_ref[1]) and st[2:] == st_ref[2:]
        assert max(seen) >= 3 and len(seen) >= 3, seen     # the multi-pass code really ran
    finally:
        lib.pbbi_host_debug_set_pass(1.02, 1024, 15)
        lib.pbbi_host_set_threads(threads0)


def test_fast_host_stream_in_place_draws(monkeypatch):
    """scaled_normal_into / uniform_into (the class API's per-iteration draws, written straight into the
    upload buffers) equal `standard_normal((D, N)) * pStd` and `uniform(size=N)` bit for bit, state included;
    wrong buffers are refused."""
    from physicsbasedbayesianinference_amd import _hoststream as hs
    for min_fast in (1, 1 << 40):            # the C path and the NumPy path of the same functions
        monkeypatch.setattr(hs, "MIN_FAST", min_fast)
        for D, N in ((3, 7), (128, 1001), (5, 1), (1, 40000)):
            scale = np.random.RandomState(D).uniform(0.5, 2.0, N)
            np.random.seed(11)
            np.random.standard_normal(1)       # leaves a cached variate
            ref_p, ref_u = np.random.standard_normal((D, N)) * scale, np.random.uniform(size=N)
            ref_state = np.random.get_state()
            np.random.seed(11)
            np.random.standard_normal(1)
            p, u = np.full((D, N), np.nan), np.full(N, np.nan)
            assert hs.scaled_normal_into(p, scale) is p and hs.uniform_into(u) is u
            got_state = np.random.get_state()
            assert np.array_equal(p, ref_p) and np.array_equal(u, ref_u), (D, N, min_fast)
            assert np.array_equal(ref_state[1], got_state[1]) and ref_state[2:] == got_state[2:]
    with pytest.raises(ValueError):
        hs.scaled_normal_into(np.zeros((4, 6), np.float32), np.ones(6))
    with pytest.raises(ValueError):
        hs.scaled_normal_into(np.zeros((4, 6)), np.ones(4))
    with pytest.raises(ValueError):
        hs.uniform_into(np.zeros((4, 6))[:, ::2])


def test_ensemble_draws_through_the_fast_stream_match_the_golden_run(monkeypatch):
    """Ensemble.setPosition / setMomentum on the fast stream reproduce the reference's recorded draws."""
    import physicsbasedbayesianinference_amd as P
    from physicsbasedbayesianinference_amd import _hoststream as hs
    from conftest import load_golden
    from scipy.constants import k as kB
    monkeypatch.setattr(hs, "MIN_FAST", 1)
    g = load_golden("G4_getsamples_dense_d128")
    np.random.seed(int(g["seed"]))
    ens = P.Ensemble(int(g["D"]), int(g["N"]))
    ens.mass = g["mass"].copy()
    assert np.array_equal(ens.setPosition(float(g["qStd"])), g["q0"])
    for i in range(int(g["S"])):
        assert np.array_equal(ens.setMomentum(float(g["temperature"])), g["p_draw"][i])
        assert np.array_equal(hs.uniform(int(g["N"])), g["u"][i])
