"""CPU suite: the C-ABI library loads without a GPU, exports every symbol the
header declares, and its host-only entry points behave (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _header_functions():
    src = open(os.path.join(ROOT, "include", "rtpose_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([a-z_0-9]+)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if not n.startswith("RTPOSE")))


def test_every_declared_symbol_is_exported(capi):
    names = _header_functions()
    assert len(names) >= 35 and "rtpose_conv2d" in names and "process_paf" in names
    for n in names:
        assert hasattr(capi.lib, n), "header declares %s but the library does not export it" % n
    # and the Python binding table covers the whole header
    assert set(names) == set(capi.EXPORTED)


def test_ctypes_mirrors_have_the_layout_of_the_header_structs(capi, tmp_path):
    """The descriptors of the ABI have grown by trailing fields (rtpose_conv_desc: wino_m, then the channel-plane fields):
    a binding that lags behind passes a shorter struct and the library reads garbage past it.  The header is compiled as C
    (gcc) and every struct's size and field offsets are compared with the ctypes mirror the Python side uses."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    mirrors = {"rtpose_layout": capi.Layout, "rtpose_conv_desc": capi.ConvDesc, "rtpose_pw_desc": capi.PwDesc,
               "rtpose_net_options": capi.NetOptions, "rtpose_prep_image": capi.PrepImage,
               "rtpose_decode_cfg": capi.DecodeCfg}
    cname = {"inp": "in"}                       # ctypes field -> C member where the names differ
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rtpose_mi355x.h"', 'int main(void) {']
    for st, cls in mirrors.items():
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for f, _ in cls._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, f, st, cname.get(f, f)))
    lines += ['  return 0;', '}']
    src = tmp_path / "abi_layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "abi_layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE, text=True).stdout
    got = dict(ln.split() for ln in out.splitlines())
    for st, cls in mirrors.items():
        assert int(got[st]) == C.sizeof(cls), (st, got[st], C.sizeof(cls))
        for f, _ in cls._fields_:
            assert int(got["%s.%s" % (st, f)]) == getattr(cls, f).offset, (st, f)


def test_version_and_sizes(capi):
    lib = capi.lib
    assert b"gfx950" in lib.rtpose_version()
    # packed weights: (k*k * cin rounded to 8 + two taps of prefetch slack) * cout rounded to 64
    assert lib.rtpose_packed_weight_floats(38, 185, 7) == (49 * 192 + 32) * 64
    assert lib.rtpose_packed_weight_floats(64, 3, 3) == (9 * 8 + 32) * 64
    assert lib.rtpose_packed_bias_floats(19) == 64
    lay = capi.Layout.padded(192, 46, 46, 3)
    assert (lay.ws, lay.hs, lay.lead) == (49, 49, 3 * 49 + 3)
    assert lib.rtpose_layout_pixels(C.byref(lay), 32, 46, 46) >= lay.lead + 32 * 49 * 49 + 3


def test_net_plan_introspection_matches_reference_state_dict(capi, pkg):
    lib = capi.lib
    h = C.c_void_p()
    capi.check(lib.rtpose_net_create(2, 368, 368, C.byref(h)))
    try:
        n = lib.rtpose_net_num_convs(h)
        assert n == 92                                        # SURVEY: 92 nn.Conv2d
        m = pkg.get_model('vgg19')
        convs = m._convs()
        name = C.create_string_buffer(64)
        co, ci, k = C.c_int(), C.c_int(), C.c_int()
        total = 0
        for i, (nm, mod) in enumerate(convs):
            capi.check(lib.rtpose_net_conv_info(h, i, name, 64, C.byref(co), C.byref(ci), C.byref(k)))
            assert name.value.decode() == nm
            assert tuple(mod.weight.shape) == (co.value, ci.value, k.value, k.value)
            total += mod.weight.numel() + mod.bias.numel()
        assert total == 52311446                              # SURVEY §6 parameter count
        assert lib.rtpose_net_workspace_bytes(h) > 0 and lib.rtpose_net_weight_bytes(h) > total * 4
        # algorithmic flops of the plan == SURVEY's 271.868 GFLOP/img
        fl = 0.0
        f, kk = C.c_double(), C.c_int()
        for i in range(lib.rtpose_net_num_launches(h)):
            capi.check(lib.rtpose_net_launch_info(h, i, None, C.byref(kk), C.byref(f), None, 0))
            fl += f.value
        assert abs(fl / 2 / 1e9 - 271.868) < 0.01
        # forward before bind is a loud state error, not a crash
        assert lib.rtpose_net_forward(h, C.c_void_p(16), None) != 0
        assert "not bound" in capi.last_error()
    finally:
        lib.rtpose_net_destroy(h)


def test_bad_arguments_are_rejected_on_the_host(capi):
    lib = capi.lib
    h = C.c_void_p()
    assert lib.rtpose_net_create(0, 368, 368, C.byref(h)) != 0
    cfg = capi.DecodeCfg(18, 8, 0.1, 100000, 64)
    assert lib.rtpose_decode_result_bytes(C.byref(cfg), 4) == 0          # capacity above the limit
    cfg = capi.DecodeCfg(18, 8, 0.1, 32, 64)
    assert lib.rtpose_decode_result_bytes(C.byref(cfg), 4) == 4 * 4 * (32 + 72 * 32 + 19 * 64)
    assert lib.rtpose_decode_workspace_bytes(C.byref(cfg), 4) >= 4 * 19 * (1 + 3 * 32) * 4
    # legacy getters on an empty state: bounds-checked, never UB
    assert lib.get_part_cid(3, 3) == -1 and lib.get_part_x(0) == -1


def test_compute_dtype_plans_on_the_host(capi, pkg):
    """Plan construction is host-only: the reduced-precision plans exist, report their dtype, keep
    the same 92 convs / launch list, size their arenas sensibly and reject what they cannot run."""
    lib = capi.lib
    sizes = {}
    for dt in (capi.DTYPE_F32, capi.DTYPE_BF16, capi.DTYPE_BF16X3):
        h = C.c_void_p()
        capi.check(lib.rtpose_net_create_ex(2, 368, 368, dt, C.byref(h)))
        try:
            assert lib.rtpose_net_dtype(h) == dt
            assert lib.rtpose_net_num_convs(h) == 92
            sizes[dt] = (lib.rtpose_net_workspace_bytes(h), lib.rtpose_net_weight_bytes(h), lib.rtpose_net_num_launches(h))
        finally:
            lib.rtpose_net_destroy(h)
    f32, bf16, x3 = sizes[capi.DTYPE_F32], sizes[capi.DTYPE_BF16], sizes[capi.DTYPE_BF16X3]
    assert 0.4 * f32[0] < bf16[0] < 0.7 * f32[0]        # 2-byte activations (+ fp32 staging / records)
    assert 0.9 * f32[0] < x3[0] < 1.2 * f32[0]          # hi + lo = the fp32 footprint
    # weights: bf16 = half of the plain fp32 packing = the bf16x3 one; the fp32 arena holds EVERY packing a plan may
    # choose (rtpose_net_options; the arena is shared by all plans of a module): direct (1) + F(2x2,3x3) (16/9) +
    # F(4x4,3x3) (36/9) for the 3x3 convs, direct (1) + F(4,7) (70/49) + F(6,7) (84/49) for the 7x7 convs
    assert 0.45 * x3[1] < bf16[1] < 0.6 * x3[1] and 4.0 * x3[1] < f32[1] < 4.8 * x3[1], (f32[1], x3[1])
    # ... and the workspace of an fp32 plan includes the hand-over scratch of the persistent 7x7 launches
    assert lib.rtpose_conv2d_winograd_scratch_bytes() > 0
    # bf16 / bf16x3: stage 6 writes its fp32 record directly (no save copy); fp32 and - since round 6 - bf16: the two trailing
    # 1x1 convs of all six stages run as one back-to-back launch each (csrc/conv_tail.hip, conv_tail_bf16.hip); the split plan
    # keeps them as two launches of its generic kernel
    assert bf16[2] == f32[2] - 1 and x3[2] == f32[2] - 1 + 6
    h = C.c_void_p()
    assert lib.rtpose_net_create_ex(1, 364, 368, capi.DTYPE_BF16, C.byref(h)) != 0   # not a multiple of 8
    assert "multiples of 8" in capi.last_error()
    assert lib.rtpose_net_create_ex(1, 368, 368, 7, C.byref(h)) != 0
    assert lib.rtpose_shufflenet_create_ex(1, 368, 368, capi.DTYPE_BF16X3, C.byref(h)) != 0
    # packed-weight sizes: bf16 = 2 bytes per element of the padded filter, bf16x3 twice that
    assert lib.rtpose_packed_weight_bytes_bf16(128, 128, 7) == (49 * 128 + 128) * 128 * 2
    assert lib.rtpose_packed_weight_bytes_bf16x3(128, 128, 7) == 2 * lib.rtpose_packed_weight_bytes_bf16(128, 128, 7)
    m = pkg.get_model('vgg19')
    import pytest as _pt
    with _pt.raises(ValueError):
        m.set_compute_dtype('fp16')
    assert m.set_compute_dtype('bf16x3').compute_dtype == 'bf16x3'


def test_output_guard_position_per_arithmetic_and_environment(capi, monkeypatch):
    """Where a guarded forward waits for the reader of its previous maps (rtpose_net_set_output_guard, the decoder of the
    batch before on a second stream): by default in front of the FIRST launch that writes the buffer the maps live in -
    fp32: conv4_4_CPM (`model0.25`, it writes the out1 channels of the same concat buffer), so that the reader runs beside
    the trunk; bf16 / bf16x3: the last launch (the stage-6 pair writes the fp32 record) - and in front of the whole launch
    list under RTPOSE_GUARD_WHOLE_FORWARD=1 (or RTPOSE_GUARD_FINE=0).  Host-only: rtpose_net_output_guard_launch."""
    lib = capi.lib
    name = C.create_string_buffer(96)

    def guard_launch(dt):
        h = C.c_void_p()
        capi.check(lib.rtpose_net_create_ex(2, 368, 368, dt, C.byref(h)))
        try:
            g = lib.rtpose_net_output_guard_launch(h)
            capi.check(lib.rtpose_net_launch_info(h, g, None, None, None, name, 96))
            assert lib.rtpose_net_set_output_guard(h, None) == 0 and lib.rtpose_net_output_guard_launch(h) == g
            return g, name.value.decode(), lib.rtpose_net_num_launches(h)
        finally:
            lib.rtpose_net_destroy(h)
    monkeypatch.delenv("RTPOSE_GUARD_WHOLE_FORWARD", raising=False)
    monkeypatch.delenv("RTPOSE_GUARD_FINE", raising=False)
    g, nm, n = guard_launch(capi.DTYPE_F32)
    assert nm == "model0.25" and 0 < g < n - 1
    for dt in (capi.DTYPE_BF16, capi.DTYPE_BF16X3):
        g, nm, n = guard_launch(dt)
        assert g == n - 1 and "model6_1" in nm and "model6_2" in nm
    for var, val in (("RTPOSE_GUARD_WHOLE_FORWARD", "1"), ("RTPOSE_GUARD_FINE", "0")):
        monkeypatch.setenv(var, val)
        for dt in (capi.DTYPE_F32, capi.DTYPE_BF16):
            assert guard_launch(dt)[0] == 0
        monkeypatch.delenv(var)


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/rtpose_mi355x.h must compile as C99 on its own
    (no C++, no HIP, no torch types in the signatures)."""
    import os
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        import pytest as _pt
        _pt.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "rtpose_mi355x.h"\nint main(void) { return rtpose_version() == 0; }\n')
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"),
                    "-c", str(src), "-o", str(tmp_path / "hdr.o")], check=True)


def test_winograd_fits_is_host_only_logic(capi):
    """rtpose_conv2d_winograd_fits decides on the host (csrc/conv_wino.hip, conv_wino7.hip): 3x3 needs cin % 16 (or % 8
    for 64 padded columns), 7x7 needs cin % 8, 128-column tiles and transformed rows that fit the LDS."""
    lib = capi.lib
    d = (capi.ConvDesc * 1)()

    def fits(k, cin, cout, n=1, h=46, w=46, pool=0):
        d[0].k, d[0].cin, d[0].cout, d[0].pool = k, cin, cout, pool
        d[0].lin = capi.Layout.padded(cin, h, w, k // 2)
        return lib.rtpose_conv2d_winograd_fits(d, n, h, w)

    assert fits(3, 8, 128) == 0 and fits(3, 24, 64) == 1 and fits(3, 512, 512) == 1 and fits(1, 128, 128) == 0
    assert fits(3, 8, 64) == 0 and fits(3, 16, 128) == 0 and fits(3, 32, 128) == 1    # at least two chunks
    assert fits(7, 128, 128, 32) == 1 and fits(7, 192, 128, 32) == 1 and fits(7, 128, 128, 1) == 1
    assert fits(7, 128, 38) == 0 and fits(7, 100, 128) == 0 and fits(7, 128, 128, 1, 46, 46, pool=1) == 0
    assert fits(7, 128, 128, 2, 184, 184) == 0      # transformed rows of a 184-wide map do not fit the LDS
    assert fits(7, 128, 128, 4, 69, 87) == 1        # a TTA scale: two transform items per thread
    # packed sizes: 16 / 9 of the taps for 3x3, (FM + 6) * 7 / 49 for 7x7 (F(6,7) unless RTPOSE_WINOGRAD7_M=4), + slack
    assert lib.rtpose_packed_weight_floats_winograd(128, 128, 3) == (16 * 128 + 64) * 128
    assert lib.rtpose_packed_weight_floats_winograd(128, 128, 7) in ((84 * 128 + 96) * 128, (70 * 128 + 96) * 128)
    assert lib.rtpose_packed_weight_floats_winograd7(128, 128, 6) == (84 * 128 + 96) * 128
    assert lib.rtpose_packed_weight_floats_winograd7(128, 128, 4) == (70 * 128 + 96) * 128
    # the form is part of the descriptor: F(4,7) has more position groups per row, hence fits narrower maps only
    d[0].wino_m = 4
    assert fits(7, 128, 128, 32) == 1
    d[0].wino_m = 6
    assert fits(7, 128, 128, 32) == 1
    d[0].wino_m = 5
    assert fits(7, 128, 128, 32) == 0
    d[0].wino_m = 0


def test_conv3x3_c64_fits_is_host_only_logic(capi):
    """rtpose_conv3x3_c64_bf16_fits (round 6, csrc/conv_c64_bf16.hip) decides on the host which 3x3 bf16 convs
    rtpose_conv2d_bf16 hands to the 64-input-channel kernel: conv1_2 (+ pool) and conv2_1 of the VGG-19 front end in the
    network's layouts at every size of the multi-scale flow that fits 32-bit byte offsets - and nothing else."""
    lib, Layout = capi.lib, capi.Layout

    def desc(cin, cout, h, w, pool, pin=1, cs_in=None, ch_in=0, cs_out=None, ch_out=0):
        d = (capi.ConvDesc * 1)()
        ho, wo = (h // 2, w // 2) if pool else (h, w)
        d[0].lin = Layout.padded(cs_in or cin, h, w, pin, choff=ch_in)
        d[0].lout = Layout.padded(cs_out or cout, ho, wo, 1, choff=ch_out)
        d[0].cin, d[0].cout, d[0].k, d[0].relu, d[0].pool = cin, cout, 3, 1, pool
        return d

    for n, size in ((32, 368), (32, 184), (32, 552), (32, 736), (1, 46)):
        assert lib.rtpose_conv3x3_c64_bf16_fits(desc(64, 64, size, size, 1), 1, n, size, size) == 1      # conv1_2 + pool
        assert lib.rtpose_conv3x3_c64_bf16_fits(desc(64, 128, size // 2, size // 2, 0), 1, n, size // 2, size // 2) == 1  # conv2_1
    assert lib.rtpose_conv3x3_c64_bf16_fits(desc(64, 64, 368, 368, 1, cs_in=80, ch_in=16, cs_out=80, ch_out=8), 1, 2, 368, 368) == 1
    assert lib.rtpose_conv3x3_c64_bf16_fits(desc(128, 128, 184, 184, 1), 1, 32, 184, 184) == 0     # other channel counts
    assert lib.rtpose_conv3x3_c64_bf16_fits(desc(64, 96, 184, 184, 0), 1, 32, 184, 184) == 0       # cout not a multiple of 64
    assert lib.rtpose_conv3x3_c64_bf16_fits(desc(64, 64, 368, 368, 1, cs_out=67, ch_out=1), 1, 2, 368, 368) == 0   # unaligned slice
    assert lib.rtpose_conv3x3_c64_bf16_fits(desc(64, 64, 368, 368, 1, pin=0), 1, 2, 368, 368) == 0  # no gap for the padding
    assert lib.rtpose_conv3x3_c64_bf16_fits(desc(64, 64, 367, 368, 1), 1, 2, 367, 368) == 0       # fused pool needs even sizes
    assert lib.rtpose_conv3x3_c64_bf16_fits(desc(64, 64, 368, 368, 1), 2, 32, 368, 368) == 0      # grouped launches
    assert lib.rtpose_conv3x3_c64_bf16_fits(desc(64, 64, 1104, 1104, 1), 1, 32, 1104, 1104) == 0  # 5 GB: past 32-bit offsets
    d7 = desc(64, 64, 368, 368, 0)
    d7[0].k = 7
    assert lib.rtpose_conv3x3_c64_bf16_fits(d7, 1, 2, 368, 368) == 0


def test_decoder_objects_hold_no_packed_fp32_instruction(tmp_path):
    """The decoder's kernels run on a second stream beside the forward's MFMA kernels (pipeline.SideDecoder).  Round 6
    (DESIGN.md 3.3): with clang's vectorisers on, limb_assign_kernel's sample loop was compiled into packed-fp32 VALU
    instructions (v_pk_mul_f32 / v_pk_add_f32) and returned wrong scores in lanes 48..63 beside the bf16 plan's kernels
    (~2 launches in 1000); the same source without them: 0 of 240,000.  csrc/Makefile therefore builds decode.hip /
    legacy_pafprocess.hip with -fno-slp-vectorize -fno-vectorize; this test disassembles the gfx950 code of the built
    objects and refuses any packed-fp32 arithmetic instruction in them."""
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("no ROCm LLVM tools here")
    build = os.path.join(ROOT, "pytorch_realtime_multi-person_pose_estimation_amd", "csrc", "build")
    obj = os.path.join(build, "decode.o")
    if not os.path.exists(obj):
        pytest.skip("csrc/build/decode.o not built (run __graft_entry__.build())")
    fat, co = str(tmp_path / "decode.fatbin"), str(tmp_path / "decode.co")
    subprocess.run([tools[0], "--dump-section", ".hip_fatbin=" + fat, obj], check=True)
    subprocess.run([tools[1], "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co,
                    "--unbundle"], check=True)
    dis = subprocess.run([tools[2], "-d", co], check=True, stdout=subprocess.PIPE, text=True).stdout
    assert "limb_assign_kernel" in dis and "global_load_dword" in dis        # (it is the device code we are looking at)
    packed = sorted(set(re.findall(r"\bv_pk_[a-z0-9_]*f32\b", dis)))
    assert not packed, "packed-fp32 instructions in the decoder's device code: %s" % packed
    mk = open(os.path.join(ROOT, "pytorch_realtime_multi-person_pose_estimation_amd", "csrc", "Makefile")).read()
    assert re.search(r"build/decode\.o build/legacy_pafprocess\.o: CXXFLAGS \+= -fno-slp-vectorize -fno-vectorize", mk)
