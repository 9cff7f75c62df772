"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/agents_amd.h declares (no kernel launches here)."""
import ctypes
import os
import re

from agents_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "agents_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aa_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_binding_covers_header(lib):
    assert declared_symbols() == _lib.exported_symbols()


def declared_arity():
    """{symbol: number of parameters} parsed from the header prototypes."""
    src = open(os.path.join(ROOT, "include", "agents_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(aa_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_binding_arity_matches_header(lib):
    """Every ctypes signature in agents_amd/_lib.py has as many parameters as the C prototype."""
    arity = declared_arity()
    bad = {n: (len(sig[1]), arity.get(n)) for n, sig in _lib._SIGNATURES.items()
           if arity.get(n) != len(sig[1])}
    assert not bad, f"(binding, header) parameter counts differ: {bad}"


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of the descriptor structs as gcc sees the header == the ctypes mirrors."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("gcc not available")
    fields = {"aa_gemm_desc": ("GemmDesc", ["A", "M", "a_mode", "img_pitch", "a_div", "bias",
                                            "mask_src", "force_cfg", "colsum_out", "no_dma"]),
              "aa_conv_layer_desc": ("ConvLayerDesc", ["w", "y", "KH", "act"]),
              "aa_conv_dx_desc": ("ConvDxDesc", ["dz", "dx", "n_img", "mask_kind"]),
              "aa_plane_scatter": ("PlaneScatter", ["n", "stride", "lo", "hi", "pos", "planes"]),
              "aa_grad_slabs": ("GradSlabs", ["n", "splits", "mn", "n_tail", "offset", "slab"]),
              "aa_mlp_layout": ("MlpLayout", ["n_layers", "dims", "acts", "k_off", "b_off"]),
              "aa_ppo_fused_desc": ("PpoFusedDesc", [
                  "obs", "ld_obs", "obs_dim", "D", "actions", "old_vpred", "step_type", "N", "rows",
                  "nrm_count", "nrm_eps", "nrm_clip", "params", "total", "head_off", "actor",
                  "value", "act_mean", "act_mag", "clip_eps", "denom", "adv_eps"]),
              "aa_sac_sample_tail": ("SacSampleTail", ["net", "A", "std_kind", "act_mean", "eps_in",
                                                       "seed", "arrival_dev", "logp", "save_eps"]),
              "aa_sac_dout_gen": ("SacDoutGen", ["kind", "q1", "discount", "logp", "log_alpha",
                                                 "gamma", "loss_kind", "global_batch", "loss_out",
                                                 "dlogp_out", "z", "A", "std_kind", "act_mag",
                                                 "save_eps", "daction", "ld_daction", "daction2",
                                                 "ld_daction2", "dlogp"]),
              "aa_mlp_wide_adam": ("MlpWideAdam", ["p", "m", "v", "target", "lr", "beta1",
                                                   "beta2", "eps", "tau", "step_dev",
                                                   "arrival_dev"]),
              "aa_ppo_policy_step_desc": ("PpoPolicyStepDesc", [
                  "x", "ldx", "B", "nrm_mean", "nrm_var_den", "nrm_eps", "nrm_clip", "params_a",
                  "n_layers_a", "dims_a", "b_off_a", "params_b", "n_layers_b", "b_off_b",
                  "value_out", "std_bias", "act_mag", "D", "loc", "seed", "call_counter_dev",
                  "clip_hi", "action"])}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "agents_amd.h"', 'int main(void){']
    for cname, (_, fs) in fields.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f in fs:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = dict(line.split() for line in out.strip().splitlines())
    for cname, (pyname, fs) in fields.items():
        st = getattr(_lib, pyname)
        assert int(got[cname]) == ctypes.sizeof(st), cname
        for f in fs:
            assert int(got[f"{cname}.{f}"]) == getattr(st, f).offset, f"{cname}.{f}"


def test_abi_version(lib):
    assert lib.aa_abi_version() == 21


def test_argument_validation_without_gpu(lib):
    """Entry points reject bad arguments before touching the device."""
    assert lib.aa_rb_sample_rows(None, 1, 1, 1, 1, 0, 0, None, None, None, None, None) == -22
    assert lib.aa_counter_add(None, 1, None) == -22
    d = _lib.GemmDesc()
    assert lib.aa_gemm_f32_workspace_bytes(ctypes.byref(d)) == -1
    assert lib.aa_colsum_workspace_bytes(0, 4) == -1
    assert lib.aa_colsum_workspace_bytes(1000, 32) > 0


def test_ppo_merge_apply_switch(lib):
    """aa_ppo_fused_merge_apply: reads with a negative argument, returns the previous value, and
    the workspace it sizes has room for the barrier slots of the merged launch."""
    before = lib.aa_ppo_fused_merge_apply(-1)
    try:
        assert before in (0, 1)
        assert lib.aa_ppo_fused_merge_apply(0) == before
        assert lib.aa_ppo_fused_merge_apply(-1) == 0
        assert lib.aa_ppo_fused_merge_apply(7) == 0
        assert lib.aa_ppo_fused_merge_apply(-1) == 1
    finally:
        lib.aa_ppo_fused_merge_apply(before)
    n_wg, total = 256, 11092
    floats = n_wg * total + n_wg * 8 + (total + 15) // 16 + 16 + 2 * 4096
    assert lib.aa_ppo_fused_workspace_bytes(4096, total) >= (floats + 1) * 4 + 8 * 257


def test_prio_draw_validates_on_the_host(lib):
    """aa_prio_draw_rows refuses what it cannot run before any launch: null arguments, a workspace
    that is too small or misaligned, tables of more than 8,000 blocks of 1,024 rows (those take
    the three-launch aa_prio_sample_rows)."""
    assert lib.aa_prio_draw_workspace_bytes(0) == -1
    assert lib.aa_prio_draw_workspace_bytes(1) == 3 * 8             # one slot + two control words
    assert lib.aa_prio_draw_workspace_bytes(1024 * 8000) == 8002 * 8
    assert lib.aa_prio_draw_workspace_bytes(1024 * 8000 + 1) == -1
    assert lib.aa_prio_workspace_bytes(1024 * 8000 + 1) == 8001 * 8  # the fallback still sizes it
    buf = (ctypes.c_uint64 * 64)()
    a = ctypes.addressof(buf)
    a16 = (a + 15) & ~15
    args = lambda **kw: [kw.get("pq", a16), kw.get("ids", a16), a16, kw.get("batch", 4),
                         kw.get("L", 16), kw.get("S", 8), 2, 7, a16, kw.get("ws", a16),
                         kw.get("ws_bytes", 24), a16, None, None, None, None]
    assert lib.aa_prio_draw_rows(*args(pq=None)) == -22
    assert lib.aa_prio_draw_rows(*args(S=0)) == -22
    assert lib.aa_prio_draw_rows(*args(ws_bytes=16)) == -34
    assert lib.aa_prio_draw_rows(*args(ws=a16 + 4)) == -34
    assert lib.aa_prio_draw_rows(*args(batch=1024, L=8001, ws_bytes=1 << 20)) == -34
    assert lib.aa_prio_draw_rows(*args(ids=a16 + 8)) == -22


def test_round3_entries_validate_on_the_host(lib):
    """The entry points added in round 3 reject what they cannot run before any launch: the
    wide-MLP limits (widths <= 256 behind a <= 1,024-wide input, <= 4 layers, batch <= 1,024),
    stamped replay draws, the fused SAC tail."""
    def layout(dims, acts=None):
        lay = _lib.MlpLayout()
        lay.n_layers = len(dims) - 1
        for i, d in enumerate(dims):
            lay.dims[i] = d
        for i in range(len(dims) - 1):
            lay.acts[i] = (acts or [1] * (len(dims) - 1))[i]
            lay.k_off[i] = 0
            lay.b_off[i] = 0
        return lay
    ok = layout([393, 256, 256, 1], [1, 1, 0])
    assert lib.aa_mlp_wide_supported(ctypes.byref(ok), 256) == 1
    assert lib.aa_mlp_wide_supported(ctypes.byref(ok), 1024) == 1
    assert lib.aa_mlp_wide_supported(ctypes.byref(ok), 1025) == 0          # GEMM path beyond that
    assert lib.aa_mlp_wide_supported(ctypes.byref(layout([1025, 64, 1])), 8) == 0
    assert lib.aa_mlp_wide_supported(ctypes.byref(layout([64, 257, 1])), 8) == 0
    assert lib.aa_mlp_wide_supported(ctypes.byref(layout([64, 64, 1], [1, 7])), 8) == 0
    assert lib.aa_mlp_wide_supported(None, 8) == 0
    f = _lib.MlpWideFwd()
    f.layout = ok
    f.n_nets, f.x_split, f.B = 1, 393, 256
    assert lib.aa_mlp_wide_forward(ctypes.byref(f), None) == -22            # no parameter pointer
    f.n_nets = 5
    assert lib.aa_mlp_wide_forward(ctypes.byref(f), None) == -22
    b = _lib.MlpWideBwd()
    b.layout = ok
    b.n_nets, b.x_split, b.B = 1, 376, 256
    assert lib.aa_mlp_wide_backward(ctypes.byref(b), None) == -22
    assert lib.aa_mlp_wide_backward(None, None) == -22
    # stamped draws: S, T, batch, max_len must be positive; ids need an id table
    assert lib.aa_rb_sample_gather_stamped(None, None, None, 0, None, None, None, 5, 4, 8, 0, 2, 0,
                                           0, None, None, None) == -22
    assert lib.aa_rb_sample_gather_stamped(None, None, None, 0, None, 1 << 20, None, 5, 4, 8, 2, 2,
                                           0, 0, None, None, None) == -22
    assert lib.aa_sac_alpha_step(None, None, None, 0.0, 1, 1.0, 8, 8.0, None, None, None, None,
                                 None, 1e-3, 0.9, 0.999, 1e-7, None, None, None, None) == -22
    assert lib.aa_sac_head_backward(1 << 20, 4, 2, 1 << 20, 0, 1 << 20, 1 << 20, 1 << 20, 1 << 20, 1,
                                    None, 0, 1 << 20, 1 << 20, None) == -22   # row stride < A


def test_slab_optimizer_validates_on_the_host(lib):
    """aa_rmsprop_step_slabs rejects segment tables it cannot walk before any launch."""
    P = 1 << 20

    def call(g, n=4096):
        return lib.aa_rmsprop_step_slabs(P, P, P, None, None, n, 1e-3, 0.9, 0.0, 1e-7, None,
                                         ctypes.byref(g), None)

    def seg(splits=64, mn=256, n_tail=16, offset=0, slab=P, n=1):
        g = _lib.GradSlabs()
        g.n = n
        g.splits[0], g.mn[0], g.n_tail[0], g.offset[0], g.slab[0] = splits, mn, n_tail, offset, slab
        return g
    assert call(seg(n=5)) == -22                 # more than AA_MAX_GRAD_SLABS segments
    assert call(seg(slab=None)) == -22
    assert call(seg(mn=254)) == -22              # 16-byte items
    assert call(seg(offset=2)) == -22
    assert call(seg(offset=4000)) == -22         # runs past the parameter vector
    assert call(seg(splits=8)) == -34            # not the 16-z-lane shape of the reduce launch
    assert call(seg(mn=4 * 65536 + 4), n=1 << 20) == -34
    two = seg(n=2)
    two.splits[1], two.mn[1], two.n_tail[1], two.offset[1], two.slab[1] = 64, 64, 0, 128, P
    assert call(two) == -22                      # overlapping segments


def test_shape_qualification_is_host_side(lib):
    """The per-frame conv kernels decide on the host which shapes they take (no device access):
    the Atari conv2 -> conv3 pair and both input gradients qualify, the 84x84 first layer does not;
    the GEMM planner reports workspace for the plans a shape selects."""
    def layer(kh, kw, s, cout):
        return _lib.ConvLayerDesc(w=None, bias=None, y=None, KH=kh, KW=kw, stride=s, Cout=cout, act=0)
    a, b = layer(4, 4, 2, 64), layer(3, 3, 1, 64)
    assert lib.aa_conv_pair_supported(256, 20, 20, 32, ctypes.byref(a), ctypes.byref(b)) == 1
    c = layer(8, 8, 4, 32)
    assert lib.aa_conv_pair_supported(256, 84, 84, 4, ctypes.byref(c), ctypes.byref(a)) == 0
    assert lib.aa_conv_pair_forward(None, 0, 256, 20, 20, 32, ctypes.byref(a), ctypes.byref(b),
                                    None) == -22
    def dx(n, h, w, cin, kh, kw, s, cout):
        return _lib.ConvDxDesc(dz=None, w=None, mask_src=None, dx=None, n_img=n, H=h, W=w, Cin=cin,
                               KH=kh, KW=kw, stride=s, Cout=cout, mask_kind=0)
    assert lib.aa_conv_dx_frame_supported(ctypes.byref(dx(256, 20, 20, 32, 4, 4, 2, 64))) == 1
    assert lib.aa_conv_dx_frame_supported(ctypes.byref(dx(256, 9, 9, 64, 3, 3, 1, 64))) == 1
    assert lib.aa_conv_dx_frame_supported(ctypes.byref(dx(256, 84, 84, 4, 8, 8, 4, 32))) == 0
    assert lib.aa_conv_dx_frame(ctypes.byref(dx(256, 20, 20, 32, 4, 4, 2, 64)), None) == -22
    assert lib.aa_conv_dx_frame(ctypes.byref(dx(4, 40, 40, 32, 3, 3, 1, 64)), None) == -34
    # planner: the uint8 conv1 weight gradient takes the bf16 per-frame plan (one slab per frame,
    # plus the fused bias-gradient rows); force_cfg = 9 on an ineligible shape is rejected
    g = _lib.GemmDesc(A=1 << 20, B=1 << 21, C=1 << 22, M=256, N=32, K=256 * 400, lda=0, ldb=32,
                      ldc=32, a_mode=_lib.AA_A_PATCH_T_U8, b_mode=_lib.AA_B_ROW, n_img=256, H=84,
                      W=84, Cin=4, KH=8, KW=8, stride=4, a_div=255.0, colsum_out=1 << 23)
    assert lib.aa_gemm_f32_workspace_bytes(ctypes.byref(g)) == 256 * (256 * 32 + 32) * 4
    g.force_cfg = 9
    g.N = 64
    g.ldb = g.ldc = 64
    assert lib.aa_gemm_f32_workspace_bytes(ctypes.byref(g)) == -1


def test_no_product_import_of_oracle():
    """The oracle is test infrastructure: nothing under agents_amd/ may import it."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "agents_amd")):
        for f in fs:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
