"""GPU: the C ABI used from a plain C program - no Python, no torch in the
process - checked against the C oracle (tests/c/abi_roundtrip.c)."""
import os
import signal
import subprocess

import pytest

from conftest import ORACLE, PKG, ROOT

pytestmark = pytest.mark.gpu


def test_c_abi_roundtrip_without_torch(cuda, tmp_path):
    exe = str(tmp_path / "abi_roundtrip")
    # plain gcc: the host side needs nothing but the HIP runtime API header and the two libraries
    cmd = ["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "c", "abi_roundtrip.c"),
           os.path.join(ORACLE, "fp8_oracle.c"), "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           "-L" + PKG, "-lfp8mi", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C ABI round trip: ok" in out.stdout


def test_peer_allgather_c_abi_without_torch(cuda, tmp_path):
    """include/fp8mi_peer.h from a plain C host (tests/c/peer_roundtrip.c): three forked ranks, handles exchanged through shared memory, 12 back-to-back
    gathers checked on the host, argument errors, the bounded wait - no Python, no torch.distributed in those processes."""
    exe = str(tmp_path / "peer_roundtrip")
    cmd = ["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "c", "peer_roundtrip.c"), "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           "-L" + PKG, "-lfp8mi_peer", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    # its own session: on a timeout the program AND the ranks it forked are killed (exactly the process group started here)
    proc = subprocess.Popen([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True,
                            env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    try:
        stdout, stderr = proc.communicate(timeout=180)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        proc.communicate()
        raise
    print(stdout, stderr)
    assert proc.returncode == 0, stdout + stderr
    assert "peer all-gather C round trip: ok" in stdout
