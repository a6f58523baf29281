"""CPU tests (no GPU): the C-ABI library builds, loads and exports every symbol include/centerpose_hip.h
declares; host-side helpers; the product refuses to run without a device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from centerpose_amd import hip, synth

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def built():
    ge.build()
    return hip.lib()


def _declared(header):
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", header)).read(), flags=re.S)
    return set(re.findall(r"\b(cp_[a-z0-9_]+)\s*\(", text)) - {"cp_stream_t"}


def test_library_exports_every_declared_symbol_and_nothing_else(built):
    """include/centerpose_hip.h is the product ABI; centerpose_hip_testing.h holds the one test hook (kernel selection).  Every
    declaration resolves, the binding's list is the same set, and the library's dynamic symbol table holds no other C symbol
    of its own (no tuning / debug read-back entry points in the shipped build)."""
    import subprocess

    product, hooks = _declared("centerpose_hip.h"), _declared("centerpose_hip_testing.h")
    assert product and hooks == {"cp_set_debug"} and not (product & hooks)
    declared = product | hooks
    for name in sorted(declared):
        assert hasattr(built, name), "missing export %s" % name
    assert declared == set(hip.exported_symbols())
    assert b"gfx950" in built.cp_version()
    nm = subprocess.run(["nm", "-D", "--defined-only", hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    own = {ln.split()[2] for ln in nm.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TBDR"
           and not ln.split()[2].startswith(("_Z", "__hip_", "_init", "_fini"))}
    assert own == declared, sorted(own ^ declared)


def test_last_error_is_per_thread(built):
    """cp_last_error() returns the calling thread's last message (centerpose_hip.h, Threading): a failure on another thread does
    not replace it."""
    import threading

    assert built.cp_set_default_precision(7) == -1
    mine = built.cp_last_error()
    assert b"precision" in mine
    seen = {}

    def other():
        seen["before"] = built.cp_last_error()
        h = __import__("ctypes").c_void_p()
        names = (__import__("ctypes").c_char_p * 1)(b"hm")
        classes = (__import__("ctypes").c_int * 1)(1)
        built.cp_model_create(b"resnet_18", 0, 1, names, classes, 256, __import__("ctypes").byref(h))
        seen["after"] = built.cp_last_error()

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen["before"] == b"" and b"arch" in seen["after"]
    assert built.cp_last_error() == mine


def test_kernel_variant_and_role_tables_match_the_header(built):
    """CP_NUM_KERNEL_VARIANTS / CP_NUM_ROLES of the header, the name tables inside the library and the counts the ctypes
    binding sizes its profile buffers with must agree (a new kernel variant touches all three)."""
    header = open(os.path.join(REPO, "include", "centerpose_hip.h")).read()
    nv = int(re.search(r"#define\s+CP_NUM_KERNEL_VARIANTS\s+(\d+)", header).group(1))
    nr = int(re.search(r"#define\s+CP_NUM_ROLES\s+(\d+)", header).group(1))
    names = [built.cp_kernel_variant_name(v).decode() for v in range(nv)]
    assert all(n and n != "?" for n in names) and len(set(names)) == nv, names
    assert built.cp_kernel_variant_name(nv).decode() == "?"
    roles = [built.cp_role_name(r).decode() for r in range(nr)]
    assert all(n and n != "?" for n in roles) and len(set(roles)) == nr, roles
    # the binding sizes its buffers from the library's own counts (no copies of the #defines in hip.py)
    assert built.cp_num_kernel_variants() == nv and built.cp_num_roles() == nr
    abi = int(re.search(r"#define\s+CP_ABI_VERSION\s+(\d+)", header).group(1))
    assert built.cp_abi_version() == abi == hip.ABI_VERSION


def test_argument_validation_without_gpu(built):
    import ctypes

    h = ctypes.c_void_p()
    names = (ctypes.c_char_p * 1)(b"hm")
    classes = (ctypes.c_int * 1)(1)
    assert built.cp_model_create(b"resnet_18", 0, 1, names, classes, 256, ctypes.byref(h)) == -1
    assert b"arch" in built.cp_last_error()
    assert built.cp_model_create(b"dla_34", 0, 1, names, classes, 100, ctypes.byref(h)) == -1
    # ConvGRU models: a head the reference's routing table (pose_dla_dcn.py:545-563) leaves out of the output dict is
    # refused up front (it would otherwise come back as an unwritten tensor)
    names2 = (ctypes.c_char_p * 2)(b"hm", b"hps_uncertainty")
    classes2 = (ctypes.c_int * 2)(1, 16)
    assert built.cp_model_create(b"dlav1_34", 0, 2, names2, classes2, 256, ctypes.byref(h)) == -1
    assert b"hps_uncertainty" in built.cp_last_error()
    assert built.cp_model_create(b"dlav1_34", 1, 2, names2, classes2, 256, ctypes.byref(h)) == 0
    built.cp_model_destroy(h)
    assert built.cp_model_create(b"dla_34", 0, 2, names2, classes2, 256, ctypes.byref(h)) == 0   # no routing without the GRU
    built.cp_model_destroy(h)
    # sizes are pure host arithmetic
    assert built.cp_decode_workspace_bytes(32, 100) >= 32 * 9 * 100 * 8
    assert built.cp_pnp_workspace_bytes(10) >= 10 * 288 * 8
    assert built.cp_conv2d_workspace_bytes(64, 64, 3, 3) >= 576 * 64 * 4


def test_no_cpu_fallback(built):
    x = torch.zeros(1, 16, 4, 4)
    with pytest.raises(RuntimeError):
        hip.dcn_v2_forward(x, torch.zeros(64, 16, 3, 3), torch.zeros(64), torch.zeros(1, 18, 4, 4),
                           torch.zeros(1, 9, 4, 4), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):
        hip.decode_raw(torch.zeros(1, 1, 16, 16), torch.zeros(1, 16, 16, 16), torch.zeros(1, 2, 16, 16),
                       torch.zeros(1, 8, 16, 16))


def test_synth_weights_are_deterministic_and_calibrated():
    a = synth.make_state_dict("dlav1_34")
    b = synth.make_state_dict("dlav1_34")
    for k in a:
        assert torch.equal(a[k], b[k])
    assert float(a["hm.3.bias"][0]) == pytest.approx(-2.19)
    assert synth.load_scales("dla_34", False), "synth_scales.json missing"
    x = synth.frames(2, seed=1, h=32, w=32)
    assert x.shape == (2, 3, 32, 32) and x.dtype == torch.float32
    assert abs(float(x.mean())) < 0.5


def test_detection_record_layout_matches_reference_keys():
    keys = ["bboxes", "scores", "kps", "clses", "obj_scale", "obj_scale_uncertainty", "tracking", "tracking_hp",
            "kps_displacement_mean", "kps_displacement_std", "kps_heatmap_mean", "kps_heatmap_std",
            "kps_heatmap_height"]  # decode.py:347-361
    assert list(hip.DET_FIELDS) == keys
    off = 0
    for k, (o, w) in hip.DET_FIELDS.items():
        assert o == off
        off += w
    assert off == hip.DET_STRIDE == 118


def test_device_code_has_no_packed_op_with_a_set_op_sel_bit(tmp_path):
    """Guard for the hardware hazard round 5 pinned down (tools/probe/pk_opsel_lds_hazard.hip, profiles/NOTES.md): on gfx950 a
    packed-f32 VALU op whose LOW result takes the HIGH word of a source (`op_sel:[..1..]`) returns wrong low results in lanes
    48-63 while another wave of the SIMD issues MFMAs + LDS reads -- no wait state cures it.  The library is built with
    -fno-slp-vectorize so that hipcc never makes such an op out of scalar float arithmetic; this test does not trust the flag:
    it disassembles every code object of the built library and fails on any `v_pk_*` instruction with a set op_sel bit
    (`op_sel_hi`, the broadcast form dcn16.hip uses, is measured clean)."""
    import re
    import shutil
    import subprocess
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    so = tmp_path / "lib.so"
    shutil.copy(hip.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, capture_output=True)
    objs = sorted(p for p in tmp_path.iterdir() if "amdgcn" in p.name)
    assert len(objs) >= 10, objs   # one code object per .hip source
    n_insts, bad = 0, []
    pat = re.compile(r"\bop_sel:\[[01,]*1")
    for o in objs:
        dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(o)], check=True, capture_output=True, text=True).stdout
        for line in dis.splitlines():
            if "v_pk_" in line:
                n_insts += 1
                if pat.search(line):
                    bad.append(line.strip())
            n_insts += 0
    assert n_insts > 0          # the disassembly worked (dcn16.hip's broadcast FMAs are packed ops)
    assert not bad, bad[:8]


def test_streaming_kernels_keep_their_prefetch_lead_in_the_built_library(tmp_path):
    """The row loops of the barrier-free streaming kernels (lowc.hip: lowc2_kernel, lowc1s_kernel, strm16.hip: strm16_kernel) request
    input rows two to four rows ahead.  hipcc's wait-count pass silently removes that lead -- `s_waitcnt vmcnt(0)` at the loop head -- as soon
    as the loop holds a branch around a load or store, a waterfall loop (lane-variant soffset), a register copy between prefetch
    buffers, or when the scheduler re-orders the prologue's requests (profiles/NOTES.md, round 6: every one of these happened).
    Nothing fails when it does; the kernel is just 1.4 - 4 x slower.  So: disassemble the built library, find each kernel's loops
    (backward branches) that hold its MFMAs and its row requests, and fail on any `vmcnt(0)` inside."""
    import re
    import shutil
    import subprocess
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    so = tmp_path / "lib.so"
    shutil.copy(hip.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, capture_output=True)
    # kernel -> (MFMAs, buffer loads) its row loop holds at least
    want = {"strm16_kernel": (100, 9), "lowc2_kernel": (30, 3), "lowc1s_kernel": (50, 20)}
    found = {}
    head = re.compile(r"^[0-9a-f]+ <(\S+)>:$")
    inst = re.compile(r"^\s+(\S+)(.*?)//\s*([0-9A-Fa-f]+):")
    tgt = re.compile(r"<\S+\+0x([0-9a-fA-F]+)>")
    for o in sorted(p for p in tmp_path.iterdir() if "amdgcn" in p.name):
        dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(o)], check=True, capture_output=True, text=True).stdout
        fn, start, body = None, 0, []
        funcs = {}
        for line in dis.splitlines():
            m = head.match(line)
            if m:
                fn, body = m.group(1), []
                funcs[fn] = body
                start = int(line.split()[0], 16)
                continue
            m = inst.match(line)
            if m and fn:
                t = tgt.search(line)
                body.append((int(m.group(3), 16), m.group(1), m.group(2), start + int(t.group(1), 16) if t and m.group(1).startswith(("s_cbranch", "s_branch")) else None))
        for name, body in funcs.items():
            key = next((k for k in want if k in name), None)
            if not key:
                continue
            loops = [(t, a) for a, op, _, t in body if t is not None and t <= a]
            for lo, hi in loops:
                ins = [(op, rest) for a, op, rest, _ in body if lo <= a <= hi]
                n_mfma = sum(op.startswith("v_mfma") for op, _ in ins)
                n_ld = sum(op.startswith("buffer_load") for op, _ in ins)
                if n_mfma >= want[key][0] and n_ld >= want[key][1]:
                    drains = [op + rest for op, rest in ins if op == "s_waitcnt" and "vmcnt(0)" in rest]
                    counted = [rest for op, rest in ins if op == "s_waitcnt" and "vmcnt(" in rest and "vmcnt(0)" not in rest]
                    found.setdefault(key, []).append((n_mfma, n_ld, len(counted), drains))
    for key in want:
        assert key in found, (key, "row loop not found in the disassembly", sorted(found))
        for n_mfma, n_ld, n_counted, drains in found[key]:
            assert n_counted >= 1 and not drains, (key, n_mfma, n_ld, n_counted, drains[:4])
