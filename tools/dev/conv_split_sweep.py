"""Dev probe: the few-row forward / data-gradient convolutions of R(2+1)D-18's layers 3 - 4 (frametransformer shapes) under a
forced split-K count (needs a build with the DVT_FORCE_CONV_SPLIT switch of tools/dev/force_conv_cfg.patch in conv_fwd_split)."""
import os, subprocess, sys, json
SHAPES = [  # name, frames, Cin, H, W, Cout, k, pad, stride
    ("L3 spatial 256->576", 84, 256, 14, 14, 576, (3, 3), (1, 1), 1),
    ("L3 spatial dgrad 576->256", 84, 576, 14, 14, 256, (3, 3), (1, 1), 1),
    ("L3 temporal 576->256", 28, 576, 3, 196, 256, (3, 1), (1, 0), 1),
    ("L3 temporal dgrad 256->576", 28, 256, 3, 196, 576, (3, 1), (1, 0), 1),
    ("L3 spatial dgrad 512->256", 84, 512, 14, 14, 256, (3, 3), (1, 1), 1),
    ("L4 spatial 512->1152", 56, 512, 7, 7, 1152, (3, 3), (1, 1), 1),
    ("L4 spatial dgrad 1152->512", 56, 1152, 7, 7, 512, (3, 3), (1, 1), 1),
    ("L4 spatial dgrad 960->512", 56, 960, 7, 7, 512, (3, 3), (1, 1), 1),
    ("L4 temporal 1152->512", 28, 1152, 2, 49, 512, (3, 1), (1, 0), 1),
    ("L4 temporal dgrad 512->1152", 28, 512, 2, 49, 1152, (3, 1), (1, 0), 1),
    ("L4 temporal 960->512", 28, 960, 2, 49, 512, (3, 1), (1, 0), 1),
    ("L4 temporal dgrad 512->960", 28, 512, 2, 49, 960, (3, 1), (1, 0), 1),
    ("L4 spatial/2 256->960", 84, 256, 14, 14, 960, (3, 3), (1, 1), 2),
]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
    import dvt_amd
    from dvt_amd import ops
    out = {}
    for name, N, Cin, H, W, Cout, k, pad, st in SHAPES:
        x = torch.randn(N * H * W, Cin, device="cuda").to(torch.bfloat16)
        w = torch.randn(Cout, Cin, k[0], k[1], device="cuda") * 0.03
        wp = ops.conv_weight_pack(w, ops.conv2d_implicit_k(Cin, Cout, k), torch.bfloat16)
        try:
            f = lambda: ops.conv2d_implicit(x, wp, N, Cin, H, W, Cout, k, st, pad, want_stats=True)
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(3):
                e0.record()
                for _ in range(10): f()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10 * 1e3)
            out[name] = sorted(ts)[1]
        except Exception as e:
            out[name] = None
    print("RESULT " + json.dumps(out))
    sys.exit(0)
res = {}
for sp in ["product", "1", "2", "3", "4", "6"]:
    env = dict(os.environ)
    if sp != "product":
        env["DVT_FORCE_CONV_SPLIT"] = sp
    else:
        env.pop("DVT_FORCE_CONV_SPLIT", None)
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=400)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    res[sp] = json.loads(line[0][7:]) if line else {}
    if not line:
        print(sp, "failed:", r.stderr[-500:])
print("%-30s" % "shape (kernel + reduce)" + "".join("%10s" % c for c in res))
for name, *_ in SHAPES:
    print("%-30s" % name + "".join("%10s" % ("-" if res[c].get(name) is None else "%.1f" % res[c][name]) for c in res))
