"""CPU: what the compiler made of the hot kernels, read from the code-object metadata of the shipped library (no GPU needed:
``llvm-objdump --offloading`` unbundles the gfx950 code objects, ``llvm-readelf --notes`` prints the kernel descriptors' notes).

A hand-scheduled LDS-DMA kernel that spills pays twice: a scratch reload is a compiler-placed ``s_waitcnt vmcnt(0)``, which drains
the DMA ring the counted waits exist to keep full.  So: no private segment and no spilled vector register in
any kernel of the step's hot path, every kernel inside the register budget its launch bounds promise (two waves per SIMD for the
512-thread kernels), and the one known exception — the 128-channel ping-pong kernel parks one to three registers in scratch
OUTSIDE its main loop — is held at its measured size so that it cannot grow unnoticed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "medical-image-segmentation-and-classification_amd", "mi355", "libmi355conv.so")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not (os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))):
        pytest.skip("ROCm's llvm-objdump / llvm-readelf are not installed here")
    assert os.path.exists(SO), "libmi355conv.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    d = tmp_path_factory.mktemp("co")
    so = shutil.copy(SO, d)                       # (the tool writes the unbundled code objects next to its input)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=d)
    out = {}
    for f in sorted(os.listdir(d)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, f)], check=True, capture_output=True,
                               text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            out[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
                         for k in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                   "max_flat_workgroup_size", "group_segment_fixed_size")}
    assert len(out) > 150, len(out)
    out["__dir__"] = str(d)
    return out


def _pick(kernels, pattern):
    hit = {k: v for k, v in kernels.items() if k != "__dir__" and re.search(pattern, k)}
    assert hit, pattern
    return hit


HOT = [  # (mangled-name pattern, what it is)
    (r"wgrad3x3_halo8_kernel", "eight-wave nine-tap weight gradient"),
    (r"wgrad3x3_halo_kernel", "four-wave nine-tap weight gradient"),
    (r"conv3x3_ws_kernel", "weight-stationary 3x3 (Ci = 64 / 128)"),
    (r"conv3x3_halo_rw_kernel", "four-wave halo 3x3"),
    (r"conv1x1_stream_kernel", "streaming pointwise convolution"),
    (r"conv_igemm_dma_kernel", "LDS-DMA ring implicit GEMM"),
    (r"conv_gemm256_kernel", "padding-free convolutions as 256 x 128-tile GEMMs"),
    (r"rowred_kernel|rowmap_kernel|bn_act_pool2_kernel", "BatchNorm passes"),
    (r"gate_psi_fwd|gate_bn_bwd|gate_mul", "attention-gate passes"),
    (r"adamw|bce_logits|sumsq", "loss / optimiser"),
]


@pytest.mark.parametrize("pattern,what", HOT)
def test_hot_kernels_use_no_scratch(kernels, pattern, what):
    for name, k in _pick(kernels, pattern).items():
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, (what, name, k)
        waves = -(-k["max_flat_workgroup_size"] // 64)
        if waves >= 8:                               # a 512-thread workgroup is two waves per SIMD: 256 registers each
            assert k["vgpr_count"] <= 256, (name, k)


def test_ping_pong_kernel_scratch_stays_at_its_known_size(kernels):
    """conv3x3_halo_pp128_kernel: 256 registers, 1 (bf16) / 3 (fp16) of them spilled in the prologue / epilogue, none inside the K loop
    (three ways of giving the epilogue its own lane registers made it 2, 6 and 27-31)."""
    hit = _pick(kernels, r"conv3x3_halo_pp128_kernel")
    assert len(hit) == 2
    for name, k in hit.items():
        assert k["vgpr_count"] == 256 and k["vgpr_spill_count"] <= 3 and k["private_segment_fixed_size"] <= 16, (name, k)


def test_eight_wave_weight_gradient_fits_two_waves_per_simd(kernels):
    for name, k in _pick(kernels, r"wgrad3x3_halo8_kernel").items():
        assert k["max_flat_workgroup_size"] == 512 and k["vgpr_count"] <= 256, (name, k)
    assert len(_pick(kernels, r"wgrad3x3_halo8_kernel")) == 4          # bf16 / fp16 x (64-pixel segments, 32-pixel image pairs)


STREAMING = (r"rowred_kernel|rowmap_kernel|bn_act_pool2_kernel|pack_im2col3_kernel|pack_weight_batched_kernel|pack_nchw_kernel|"
             r"gate_psi_fwd_kernel|gate_mul_bwd_kernel|wgrad_reduce_kernel")


def test_streaming_kernels_do_not_wait_for_a_load_on_the_spot(kernels):
    """The guarded-load trap ("Round 4, second half"): hipcc branches around a load that sits under a run-time
    condition and waits for it right there (s_waitcnt vmcnt(0)), which turns a loop written to keep N loads in flight into N dependent
    round trips.  Read from the disassembly of the shipped code objects: in every HBM-bound kernel, at most two loads are followed
    within two instructions by a full wait (a remainder loop's single row); the round-3 library had 8 in the BatchNorm backward apply
    pass, 4-8 in the reduce passes, 27 in the stem's im2col pack."""
    d = kernels["__dir__"]
    bad, seen = [], 0
    for f in sorted(os.listdir(d)):
        if "amdgcn" not in f:
            continue
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(d, f)], check=True, capture_output=True, text=True).stdout
        cur, n_load, n_trap, last, idx = None, 0, 0, -10, 0

        def close():
            nonlocal seen
            if cur is not None and re.search(STREAMING, cur):
                seen += 1
                if n_trap > 2:
                    bad.append((cur, n_trap, n_load))
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                close()
                cur, n_load, n_trap, last, idx = m.group(1), 0, 0, -10, 0
                continue
            t = line.strip().split()
            if cur is None or not line.startswith("\t") or not t:
                continue
            idx += 1
            if t[0].startswith(("global_load", "buffer_load")):
                n_load += 1
                last = idx
            elif t[0] == "s_waitcnt" and "vmcnt(0)" in line and idx - last <= 2:
                n_trap += 1
        close()
    assert seen >= 100, seen                     # the pattern still finds the kernels it is about
    assert not bad, bad
