"""
World > 1 through the C ABI's own collectives (rsf_comm_init / rsf_comm_init_all, rsf_pool_allgather[_all],
rsf_pool_allreduce_sum[_all]) without an 8-GPU node: tests/multirank_driver.py runs 2, 4 and 8 ctxs in one process with
the RCCL stand-in tests/c/fake_rccl.c bound through RSF_RCCL_LIB — against the CPU checker here, against the HIP library
(all ctxs on the one GPU, host and device buffers) under -m gpu.  What executes is the product's own host code around
the collectives: staging of bytes*world, the rank-major receive layout, thread-blocking init, destroy → re-init.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.fixture(scope="session")
def fake_rccl(tmp_path_factory):
    out = tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so"
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", "-o", str(out),
                           os.path.join(ROOT, "tests", "c", "fake_rccl.c"), "-lpthread", "-ldl"])
    return str(out)


def run_driver(fake, *args, hip=False):
    env = dict(os.environ, RSF_RCCL_LIB=fake, FAKE_RCCL_HIP="1" if hip else "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multirank_driver.py"), *args], env=env, capture_output=True,
                       text=True, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert lines, f"driver printed no result (status {p.returncode}):\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}"
    res = json.loads(lines[-1])
    assert p.returncode == 0 and not res["failures"], res
    return res


@pytest.mark.parametrize("params", [1, 3])
def test_checker_pools_2_4_8_ranks(fake_rccl, oracle_mod, params):
    res = run_driver(fake_rccl, "--lib", "oracle", "--params", str(params), "--chains", "192")
    assert res["checks"] > 100


def test_plain_c_caller_drives_two_ctxs(fake_rccl, oracle_mod, tmp_path):
    """tests/c/abi_smoke.c's two-ctx leg (rsf_comm_init_all from a single-threaded C program) against the checker."""
    exe = tmp_path / "abi_smoke_cpu2"
    odir = os.path.dirname(oracle_mod.lib_path())
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + odir, "-lrsf_oracle", "-lm", "-Wl,-rpath," + odir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True, env=dict(os.environ, RSF_RCCL_LIB=fake_rccl)).stdout.split("\n")
    assert out[4] == "pool ok" and out[5] == "pool2 ok", out
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True,
                         env={k: v for k, v in os.environ.items() if k != "RSF_RCCL_LIB"}).stdout.split("\n")
    assert out[5] == "pool2 skipped"


@pytest.mark.gpu
@pytest.mark.parametrize("mem,params", [("host", 1), ("device", 1), ("host", 3), ("device", 3)])
def test_hip_library_pools_2_4_8_ranks_on_one_gpu(fake_rccl, mem, params):
    res = run_driver(fake_rccl, "--lib", "hip", "--mem", mem, "--params", str(params), hip=True)
    assert res["checks"] > 100


@pytest.mark.gpu
def test_plain_c_caller_drives_two_ctxs_on_the_gpu(fake_rccl, pkg, tmp_path):
    exe = tmp_path / "abi_smoke_gpu2"
    cdir = os.path.dirname(pkg._abi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + cdir, "-lrsf_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + cdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True,
                         env=dict(os.environ, RSF_RCCL_LIB=fake_rccl, FAKE_RCCL_HIP="1")).stdout.split("\n")
    assert out[3].startswith("backend hip-gfx950") and "pool ok" in out and "pool2 ok" in out, out


@pytest.mark.gpu
@pytest.mark.parametrize("label,chains,nsteps,params,iters,sample", [("configs[3]", 2097152, 500, 1, 3, 32768), ("configs[4]", 1048576, 4000, 3, 2, 8192)])
def test_full_chain_counts_of_the_eight_gpu_configs_on_one_gpu(fake_rccl, label, chains, nsteps, params, iters, sample):
    """BASELINE configs[3] (2 097 152 chains, nsteps 500) and configs[4] (1 048 576 chains, joint (Dc, a, b), nsteps 4000) at
    their FULL chain counts: eight ctxs with the eight chain_offsets a node's ranks would have (262 144 / 131 072 chains
    each, device buffers), pooled through rsf_comm_init + rsf_pool_allgather (thread per rank, twice) and rsf_comm_init_all
    + the grouped calls (one thread) with the RCCL stand-in, every receive buffer bit-equal to ONE ctx holding all the
    chains; and 8 x `sample` of those chains, taken from every shard, against the CPU checker under the same global ids.
    Both configurations fit one GPU's HBM many times over; what the 8-GPU node adds is RCCL's transport and seven more
    devices, not a different computation."""
    res = run_driver(fake_rccl, "--lib", "hip", "--mem", "device", "--params", str(params), "--chains", str(chains), "--nsteps", str(nsteps),
                     "--iters", str(iters), "--worlds", "8", "--oracle-sample", str(sample), hip=True)
    assert res["checks"] > 150, res
