"""Dev probe (GPU): relative deviation of the 16-bit residual-stream gradient from the fp32 one at every block boundary of
the space transformer (BASELINE configs[1] shape, one clip).  No oracle involved: both runs are the HIP path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dvt_amd  # noqa
from dvt_amd import functional as F
from dvt_amd.models import vit as V
from tests.util import golden, digest_inputs, fill_state_from_numpy, rel_l2


def run(dtype, tag, prune):
    g = golden(f"vivit_{tag}_digest.npz")
    cfg, x, y = digest_inputs(g)
    net = V.ViViT(cfg["image"], cfg["patch"], cfg["classes"], cfg["frames"], dim=cfg["dim"], depth=cfg["depth"],
                  heads=cfg["heads"], dim_head=cfg["dim_head"], compute_dtype=dtype)
    fill_state_from_numpy(net.named_parameters(), int(g["fill_seed"]))
    net = net.cuda()
    grads, acts = {}, {}

    def keep(name, t):
        acts[name] = t.detach().float()
        t.register_hook(lambda gr, k=name: grads.__setitem__(k, gr.detach().float().clone()))

    st = net.space_transformer

    def fl(xx):
        keep("in", xx)
        for i, (a, f) in enumerate(st.layers):
            xx = a.fn(xx, _norm=a.norm, _residual=True); keep(f"a{i}", xx)
            xx = f.fn(xx, _norm=f.norm, _residual=True); keep(f"m{i}", xx)
        return xx

    st.forward_layers = fl
    if not prune:
        st.cls_prunable = lambda: False
    else:
        orig = st.forward_layers_cls

        def flc(xx):
            keep("in", xx)
            *head, (attn, ff) = st.layers
            for i, (a, f) in enumerate(head):
                xx = a.fn(xx, _norm=a.norm, _residual=True); keep(f"a{i}", xx)
                xx = f.fn(xx, _norm=f.norm, _residual=True); keep(f"m{i}", xx)
            an, af = attn.norm, attn.fn
            c = F.attn_block_cls(xx, an.weight, an.bias, af.to_qkv.weight, af.to_out[0].weight, af.to_out[0].bias, af.heads, eps=an.eps)
            keep("a_last_cls", c)
            c = ff.fn(c, _norm=ff.norm, _residual=True); keep("m_last_cls", c)
            return c
        st.forward_layers_cls = flc
    scale = 8192.0 if dtype == torch.float16 else 1.0
    loss = F.bce_with_logits(net(x.cuda()), y.cuda())
    loss.backward(torch.tensor(scale, device="cuda"))
    return {k: v / scale for k, v in grads.items()}, acts, {k: p.grad.float() / scale for k, p in net.named_parameters()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "c2"
    for prune in (True, False):
        g32, a32, p32 = run(torch.float32, tag, prune)
        for dt in (torch.bfloat16, torch.float16):
            g16, a16, p16 = run(dt, tag, prune)
            print(f"--- {tag} {dt} prune={prune}")
            for k in g32:
                e_all = rel_l2(g16[k], g32[k])
                if g32[k].dim() == 3:
                    e_cls = rel_l2(g16[k][:, 0], g32[k][:, 0]); e_pat = rel_l2(g16[k][:, 1:], g32[k][:, 1:])
                    n_cls, n_pat = float(g32[k][:, 0].norm()), float(g32[k][:, 1:].norm())
                    print(f"  d{k:11s} all {e_all:.2e}  cls rows {e_cls:.2e} (|g| {n_cls:.2e})  patch rows {e_pat:.2e} (|g| {n_pat:.2e})"
                          f"   act rel {rel_l2(a16[k], a32[k]):.2e}")
                else:
                    print(f"  d{k:11s} all {e_all:.2e}   act rel {rel_l2(a16[k], a32[k]):.2e}")
            print("  pos_embedding grad", f"{rel_l2(p16['pos_embedding'], p32['pos_embedding']):.2e}",
                  " space_token", f"{rel_l2(p16['space_token'], p32['space_token']):.2e}")


main()
