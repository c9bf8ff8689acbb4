"""Dev: per-workgroup s_memtime timeline of the one-pass attention backward (attn_bwd_fused_kernel).

Here (no GPU):   python tools/dev/attn_bwd_timeline.py build     -> tools/_bin/libdvt_hip_bwdtl.so: the product objects with a
                 copy of attention.hip in which the `// phase: <name>` comments of the kernel are s_memtime stamps into the
                 backward workspace (wave 0 and the dQ wave of every workgroup).
On the GPU box:  python tools/dev/attn_bwd_timeline.py run       -> phase durations (ticks of the 100 MHz s_memtime counter
                 x 21 ~ shader clocks at 2.1 GHz), averaged over the workgroups of a B*T = 256, H = 8, N = 197 launch.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "data-efficient-video-transformers_amd")
BIN = os.path.join(ROOT, "tools", "_bin")
SO = os.path.join(BIN, "libdvt_hip_bwdtl.so")
NSLOT = 32


def build():
    os.makedirs(BIN, exist_ok=True)
    subprocess.check_call([sys.executable, os.path.join(PKG, "build.py")])
    src = open(os.path.join(PKG, "csrc", "attention.hip")).read()
    k0 = src.index("__global__ __launch_bounds__(512) void attn_bwd_fused_kernel")
    k1 = src.index("// ======================================================================= host")
    body = src[k0:k1]
    counter = [0]

    def stamp(m):
        name = m.group(1).strip()
        i = counter[0]
        counter[0] += 1
        # steps / dq units repeat: slot = base + qp
        if name == "step":
            return f"if (lane == 0 && wid == 0) tlb[2 + qp] = __builtin_amdgcn_s_memtime();"
        if name == "work":
            return f"if (lane == 0 && wid == 0) tlb[24 + qp] = __builtin_amdgcn_s_memtime();"
        if name == "dq read":
            return 'asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) tlb[13] = __builtin_amdgcn_s_memtime();'
        if name == "dq mfma":
            return 'asm volatile("v_mov_b32 %0, %0" : "+v"(acc[1][3][3])); if (lane == 0) tlb[14] = __builtin_amdgcn_s_memtime();'
        if name == "dq unit":
            return f"if (lane == 0 && wid == WC) tlb[16 + qp] = __builtin_amdgcn_s_memtime();"
        slot = {"start": 0, "staged": 1, "stored": 12}[name]
        cond = "lane == 0 && wid == 0"
        return f"if ({cond}) tlb[{slot}] = __builtin_amdgcn_s_memtime();"

    body = re.sub(r"// phase: ([a-z ]+)", stamp, body)
    body = body.replace("extern __shared__ __attribute__((aligned(16))) char smem[];",
                        "extern __shared__ __attribute__((aligned(16))) char smem[];\n"
                        f"  long long* tlb = reinterpret_cast<long long*>(p.delta) + (int64_t)blockIdx.x * {NSLOT};", 1)
    out = src[:k0] + body + src[k1:]
    # the fused path hands the workspace to the kernel and sizes it for the stamps
    out = out.replace("    const int NP = p.Lkp >> 5;\n    const int waves = NP + 1;",
                      "    p.delta = (float*)d->workspace;\n    const int NP = p.Lkp >> 5;\n    const int waves = NP + 1;", 1)
    out = out.replace("  return (size_t)d->B * (size_t)d->H * (size_t)d->Lq * sizeof(float);",
                      f"  return (size_t)d->B * (size_t)d->H * ((size_t)d->Lq * sizeof(float) + {NSLOT} * 8);", 1)
    tmp = os.path.join(PKG, "csrc", "_attention_bwdtl_dev.hip")
    open(tmp, "w").write(out)
    try:
        obj = os.path.join(BIN, "attention_bwdtl.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc",
                               "-x", "hip", "-c", tmp, "-o", obj])
    finally:
        os.remove(tmp)
    objs = [os.path.join(PKG, "csrc", "_build", f) for f in os.listdir(os.path.join(PKG, "csrc", "_build"))
            if f.endswith(".o") and f != "attention.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", SO, *objs, obj, "-ldl"])
    print("built", SO)


def run():
    sys.path.insert(0, ROOT)
    import torch
    import dvt_amd
    from dvt_amd import _lib
    _lib.LIB_PATH = SO
    _lib._lib = None
    from dvt_amd import ops
    B, H, N, dh = 256, 8, 197, 64
    qkv = (torch.randn(B, N, 3, H, dh) * 0.7).to(torch.bfloat16).cuda()
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    o = torch.empty(B, N, H, dh, dtype=torch.bfloat16, device="cuda").permute(0, 2, 1, 3)
    lse = ops.attention_fwd(q, k, v, o, dh ** -0.5)
    do = torch.randn(B, N, H, dh).to(torch.bfloat16).cuda().permute(0, 2, 1, 3)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    for _ in range(3):
        ops.attention_bwd(q, k, v, o, lse, do, dq, dk, dv, dh ** -0.5)
    torch.cuda.synchronize()
    ws = ops._ws[(str(q.device), "main", torch.cuda.current_stream().cuda_stream)]
    t = ws[: B * H * NSLOT * 8].view(torch.int64).view(B * H, NSLOT).cpu()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attention_bwd(q, k, v, o, lse, do, dq, dk, dv, dh ** -0.5)
    e1.record()
    torch.cuda.synchronize()
    print(f"launch {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
    t0 = t[:, 0:1]
    d = (t - t0).double()
    names = {1: "staged", **{2 + i: f"step {i} barrier (wave 0)" for i in range(7)}, 12: "dK/dV stored (wave 0)",
             **{16 + i: f"dq unit {i} done" for i in range(7)}, **{24 + i: f"step {i} work done (wave 0)" for i in range(7)}}
    for i in sorted(names):
        print(f"{names[i]:28s} {d[:, i].mean():9.0f} ticks (+{(d[:, i] - d[:, {1: 0, 16: 1, 24: 1}.get(i, (i - 23) if i > 24 else i - 1)]).mean():7.0f})")
    print(f"last dq unit: barrier -> strip fragments landed {(t[:, 13] - t[:, 8]).double().mean():.0f}, -> MFMAs issued {(t[:, 14] - t[:, 13]).double().mean():.0f}, "
          f"-> stores issued {(t[:, 22] - t[:, 14]).double().mean():.0f}")
    life = (torch.maximum(t[:, 12], t[:, 22]) - t[:, 0]).double()
    print(f"workgroup lifetime {life.mean():.0f} ticks; span of the launch {(t[:, [12, 22]].max() - t[:, 0].min())} ticks; "
          f"8 per CU back to back = {8 * life.mean():.0f}")


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
