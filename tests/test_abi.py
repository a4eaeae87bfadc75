"""
CPU tests of the C-ABI boundary: both libraries export every symbol include/rsf_abi.h declares, the
ctypes structs match the C structs, argument validation returns error codes (no crash), and the
product library refuses to work without a GPU instead of falling back to anything.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rsf_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsf_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_declare_the_same_symbols(pkg):
    assert _declared_symbols() == sorted(pkg._abi.PROTOTYPES)


def test_oracle_library_exports_all_symbols(oracle_lib):
    for name in _declared_symbols():
        assert hasattr(oracle_lib, name), name
    assert oracle_lib.rsf_backend() == b"oracle-cpu"


def test_hip_library_loads_and_exports_all_symbols(pkg):
    """The gfx950 library is built by __graft_entry__.build(); it must load on a CPU-only host too."""
    if not os.path.exists(pkg._abi.LIB_PATH):
        pytest.skip("librsf_hip.so not built (run __graft_entry__.build())")
    lib = pkg._abi.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.rsf_backend() == b"hip-gfx950"
    assert lib.rsf_version() == pkg._abi.ABI_VERSION
    # no hard dependency on one particular libamdhip64: the binding chooses the process's single runtime
    import subprocess

    needed = subprocess.run(["objdump", "-p", pkg._abi.LIB_PATH], capture_output=True, text=True).stdout
    assert "libamdhip64" not in needed


def test_product_fails_loudly_without_gpu(pkg):
    lib = pkg._abi.load()
    if lib.rsf_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pkg._abi.RsfError):
        pkg.Engine()
    m = pkg.RateStateModel()
    m.Dc = 1000.0
    with pytest.raises(pkg._abi.RsfError):
        m.evaluate()
    with pytest.raises(pkg._abi.RsfError):
        pkg.MCMC(m, np.zeros(500), 1000.0, ["Uniform", 0.0, 1e4], 1000.0, nsamples=4).sample(False)
    # the low-level create call reports the reason through the ABI's error channel
    cfg = pkg._abi.Config()
    cfg.size, cfg.version, cfg.device = ctypes.sizeof(pkg._abi.Config), pkg._abi.ABI_VERSION, -1
    ctx = ctypes.c_void_p()
    assert lib.rsf_create(ctypes.byref(cfg), ctypes.byref(ctx)) < 0
    assert b"" != lib.rsf_last_error()


def test_struct_layouts_and_argument_validation(pkg, oracle_lib, oracle_mod):
    lib, abi = oracle_lib, pkg._abi
    ctx = ctypes.c_void_p()
    cfg = abi.Config()
    cfg.size, cfg.version = ctypes.sizeof(abi.Config) - 4, abi.ABI_VERSION
    assert lib.rsf_create(ctypes.byref(cfg), ctypes.byref(ctx)) == -1  # size mismatch is caught
    cfg.size = ctypes.sizeof(abi.Config)
    cfg.version = abi.ABI_VERSION + 1
    assert lib.rsf_create(ctypes.byref(cfg), ctypes.byref(ctx)) == -1
    cfg.version = abi.ABI_VERSION
    assert lib.rsf_create(ctypes.byref(cfg), ctypes.byref(ctx)) == 0   # => sizeof(rsf_config) agrees
    n = ctypes.c_int32()
    assert lib.rsf_model_nout(ctx, ctypes.byref(n)) == -3              # call order
    assert b"rsf_set_model" in lib.rsf_last_error()
    from bayesian_markov_chain_monte_carlo_amd.engine import _model_struct

    m = _model_struct(oracle_mod.ModelSpec(500), 1)
    assert lib.rsf_set_model(ctx, ctypes.byref(m)) == 0               # => sizeof(rsf_model) agrees
    m.substeps = 0
    assert lib.rsf_set_model(ctx, ctypes.byref(m)) == -1
    assert lib.rsf_mcmc_run(ctx, 1, None, None, None) == -3
    mc = abi.McmcConfig()
    mc.size, mc.n_params, mc.n_chains = ctypes.sizeof(abi.McmcConfig), 2, 1
    q0, data = np.array([1000.0]), np.zeros(500)
    assert lib.rsf_mcmc_init(ctx, ctypes.byref(mc), q0.ctypes.data, data.ctypes.data) == -5  # d = 2 unsupported
    mc.n_params, mc.adapt_mode, mc.adapt_interval = 3, abi.ADAPT_REFERENCE_DICT, 10
    assert lib.rsf_mcmc_init(ctx, ctypes.byref(mc), q0.ctypes.data, data.ctypes.data) == -5  # quirk mode is 1-parameter
    assert lib.rsf_destroy(ctx) == 0
    assert lib.rsf_destroy(None) == 0


def test_engine_checks_shapes(cpu_engine, oracle_mod):
    cpu_engine.set_model(oracle_mod.ModelSpec(500), 1)
    with pytest.raises(ValueError):
        cpu_engine.forward([1000.0], data=np.zeros(499), want_ssq=True)
    with pytest.raises(ValueError):
        cpu_engine.mcmc_init([[1000.0]], np.zeros(10), [0.0], [1e4])
    ssq, acc = cpu_engine.forward(np.empty(0), want_acc=True)
    assert acc.shape == (500, 0)


def test_header_compiles_as_c_and_runs_against_the_oracle(oracle_mod, tmp_path):
    """The same plain-C program the GPU suite links against librsf_hip.so (tests/c/abi_smoke.c) — here against the
    CPU oracle: the header is valid C and both libraries really share one ABI."""
    import subprocess

    exe = tmp_path / "abi_smoke_cpu"
    odir = os.path.dirname(oracle_mod.lib_path())
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + odir, "-lrsf_oracle", "-lm", "-Wl,-rpath," + odir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    assert out[0] == "nout 500" and out[3] == "backend oracle-cpu version 1 devices 0" and out[4] == "pool ok"
    ssq = np.array(out[1].split()[1:], dtype=np.float64)
    assert ssq[1] < ssq[0] and ssq[1] < ssq[2] and np.isfinite(ssq).all()
    mean, std2, acc, ev, nonfinite, done = out[2].split()[1:]
    assert 800 < float(mean) < 1200 and float(std2) > 0 and int(ev) == 640 and int(done) == 10 and int(nonfinite) == 0


def test_inconsistent_delta_t_is_refused(pkg, oracle_lib, oracle_mod):
    """The reference integrates with its delta_t attribute; a model whose delta_t no longer matches
    (t_final - t_start)/num_tsteps must not be integrated with a silently different step."""
    m = oracle_mod.ModelSpec(500)
    with pkg.Engine(lib=oracle_lib) as e:
        assert e.set_model(m, 1) == 500
        m.t_final = 40.0  # delta_t still 0.1
        with pytest.raises(ValueError, match="delta_t"):
            e.set_model(m, 1)
        m.delta_t = 40.0 / 500
        assert e.set_model(m, 1) == m.nout


def test_pool_collectives_call_order_and_single_rank(pkg, oracle_lib):
    """rsf_comm_* / rsf_pool_allgather / rsf_pool_allreduce_sum on the checker: world = 1 is the identity, anything
    before rsf_comm_init is a call-order error, and the single-process checker refuses world > 1."""
    with pkg.Engine(lib=oracle_lib) as e:
        x = np.arange(12.0).reshape(3, 4)
        with pytest.raises(pkg.RsfError, match="comm_init"):
            e.pool_allgather(x)
        assert len(e.comm_unique_id()) == 128
        with pytest.raises(pkg.RsfError, match="world = 1 only"):
            e.comm_init(2, 0, bytes(128))
        with pytest.raises(pkg.RsfError, match="bad argument"):
            e.comm_init(1, 1)
        e.comm_init(1, 0)
        with pytest.raises(pkg.RsfError, match="already has a communicator"):
            e.comm_init(1, 0)
        out = e.pool_allgather(x)
        assert out.shape == (1, 3, 4)
        np.testing.assert_array_equal(out[0], x)
        y = x.copy()
        np.testing.assert_array_equal(e.pool_allreduce_sum(y), x)
        e.comm_destroy()
        with pytest.raises(pkg.RsfError, match="comm_init"):
            e.pool_allreduce_sum(y)


def test_engine_refuses_the_checker_library_unless_declared(pkg, oracle_lib, monkeypatch):
    """Engine(lib=...) exists for the test-suite; a non-HIP library handed to it by accident must not run anything."""
    monkeypatch.delenv("RSF_ALLOW_CHECKER_ENGINE", raising=False)
    with pytest.raises(pkg.RsfError, match="no CPU fallback"):
        pkg.Engine(lib=oracle_lib)
    monkeypatch.setenv("RSF_ALLOW_CHECKER_ENGINE", "1")
    pkg.Engine(lib=oracle_lib).close()


def test_generated_trip_is_current():
    """csrc/rsf_f32_trip.inc (the float32 sampler's scheduled-assembly trip) is what tools/gen_f32_trip.py generates."""
    import subprocess
    import sys

    from conftest import ROOT

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_f32_trip.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_float32_trip_owns_its_private_register_file():
    """The float32 sampler's assembly trip keeps constants, tables and temporaries in the registers above RSF_F32_TRIP_COMPILER_VGPRS (v[154:255]) across statements.  That is
    sound only while compiled code never touches those registers, the kernel uses no AGPR, nothing spills inside the trip loop
    and the trip's 8-byte instruction stream starts on an 8-byte boundary: tools/check_private_file.py compiles one
    instantiation (no GPU) and checks all four in the ISA."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_private_file.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr[-2000:]
