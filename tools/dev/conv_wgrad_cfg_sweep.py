"""Dev probe: implicit weight gradients of R(2+1)D-18 / ResNet-18 layers 2 - 4 under every tile configuration the weight-gradient
launcher has (needs a build with tools/dev/force_conv_cfg.patch applied: the DVT_FORCE_WGRAD_CFG switch of conv_cfg)."""
import os, subprocess, sys, json
SHAPES = [  # name, frames, Cin, H, W, Cout, k, pad, stride
    ("L2 spatial 128->288", 168, 128, 28, 28, 288, (3, 3), (1, 1), 1),
    ("L2 spatial 128->256", 168, 128, 28, 28, 256, (3, 3), (1, 1), 1),
    ("L2 temporal 288->128", 28, 288, 6, 784, 128, (3, 1), (1, 0), 1),
    ("L2 temporal 256->128", 28, 256, 6, 784, 128, (3, 1), (1, 0), 1),
    ("L2 spatial/2 64->256", 336, 64, 56, 56, 256, (3, 3), (1, 1), 2),
    ("L3 spatial 256->576", 84, 256, 14, 14, 576, (3, 3), (1, 1), 1),
    ("L3 temporal 576->256", 28, 576, 3, 196, 256, (3, 1), (1, 0), 1),
    ("L3 spatial/2 128->512", 168, 128, 28, 28, 512, (3, 3), (1, 1), 2),
    ("L4 spatial 512->1152", 56, 512, 7, 7, 1152, (3, 3), (1, 1), 1),
    ("L4 temporal 1152->512", 28, 1152, 2, 49, 512, (3, 1), (1, 0), 1),
    ("R18 L2 128->128", 256, 128, 28, 28, 128, (3, 3), (1, 1), 1),
    ("R18 L2/2 64->128", 256, 64, 56, 56, 128, (3, 3), (1, 1), 2),
    ("R18 L3 256->256", 256, 256, 14, 14, 256, (3, 3), (1, 1), 1),
    ("R18 L3/2 128->256", 256, 128, 28, 28, 256, (3, 3), (1, 1), 2),
    ("R18 L4 512->512", 256, 512, 7, 7, 512, (3, 3), (1, 1), 1),
    ("R18 L4/2 256->512", 256, 256, 14, 14, 512, (3, 3), (1, 1), 2),
]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
    import dvt_amd
    from dvt_amd import ops
    out = {}
    for name, N, Cin, H, W, Cout, k, pad, st in SHAPES:
        x = torch.randn(N * H * W, Cin, device="cuda").to(torch.bfloat16)
        Ho, Wo = ops.conv_out_hw(H, W, k, st, pad)
        dz = torch.randn(N * Ho * Wo, Cout, device="cuda").to(torch.bfloat16)
        master = torch.zeros(Cout, Cin, k[0], k[1], device="cuda")
        try:
            f = lambda: ops.conv2d_implicit_wgrad(x, dz, N, Cin, H, W, Cout, k, st, pad, master=master)
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
# fail fast on a product build: without tools/dev/force_conv_cfg.patch the switch does not exist and every column would
# silently measure the product configuration
_root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
with open(os.path.join(_root, "data-efficient-video-transformers_amd", "libdvt_hip.so"), "rb") as _fh:
    if b"DVT_FORCE_WGRAD_CFG" not in _fh.read():
        sys.exit("conv_wgrad_cfg_sweep.py: libdvt_hip.so was built without tools/dev/force_conv_cfg.patch (DVT_FORCE_WGRAD_CFG is not in it)")
res = {}
for cfg in ["product", "0", "7", "1", "6"]:
    env = dict(os.environ)
    if cfg != "product":
        env["DVT_FORCE_WGRAD_CFG"] = cfg
    else:
        env.pop("DVT_FORCE_WGRAD_CFG", None)
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=400)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    res[cfg] = json.loads(line[0][7:]) if line else {}
    if not line:
        print(cfg, "failed:", r.stderr[-500:])
print("%-30s" % "shape (kernel + reduce)" + "".join("%10s" % c for c in res))
for name, *_ in SHAPES:
    print("%-30s" % name + "".join("%10s" % ("-" if res[c].get(name) is None else "%.1f" % res[c][name]) for c in res))
