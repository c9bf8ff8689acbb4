"""Dev: the small-launch zone of a vivit step from trace_steps' ordered listing (launches between the last space-transformer
forward GEMM and the first full-size weight gradient), with and without the folded CLS attention pair."""
import re
import sys
recs = []
for l in open(sys.argv[1]):
    m = re.match(r'\s*(\d+) t=\s*([\d.]+) us\s+dur=\s*([\d.]+)\s+gap=\s*([-\d.]+)\s+grid=\s*(\d+)x\s*(\d+)\s+(.*)', l)
    recs.append((int(m[1]), float(m[2]), float(m[3]), int(m[5]), m[7]))
start = next(i for i, r in enumerate(recs) if i > 20 and r[3] == 64 and 'ln_fwd' in r[4])
end = next(i for i, r in enumerate(recs) if i > start and 'gemm_dma_kernelIDF16bLb0ELb0ELi5' in r[4])
zone = recs[start:end]
cls = [r for r in zone if 'attn_cls_' in r[4]]
span = recs[end][1] - recs[start][1]
print(f"zone: {len(zone)} launches, {span:.1f} us; without the folded CLS attention pair: {len(zone) - len(cls)} launches, "
      f"{span - sum(r[2] for r in cls):.1f} us")
if len(sys.argv) > 2:
    for r in zone:
        print(f"{r[0]:4d} {r[2]:7.1f} {r[3]:5d} {r[4][:90]}")
