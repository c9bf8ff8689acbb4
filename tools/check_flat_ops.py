#!/usr/bin/env python3
"""List the kernels of the built libdvt_hip.so whose ISA contains FLAT memory instructions.

A flat_load / flat_store is what the compiler emits when it has lost a pointer's address space (an LDS buffer picked through a
run-time-indexed pointer array, a global pointer that went through a select).  On gfx950 a FLAT access counts on vmcnt AND
lgkmcnt, so a wait for an LDS fragment also waits for every outstanding global load -- the round-3 halo convolution read its
activation fragments that way for two rounds (each fragment wait also waited for the next patch's DMA).  No kernel of the
product path should contain one; tests/test_abi_cpu.py runs this check on the built library.

The milder form of the same loss: the accesses stay ds_* but their ADDRESS is a 64-bit generic pointer converted per access
(v_lshl_add_u64, v_cmp_ne_u64 against null, v_cndmask -1, per fragment read) -- the two halo weight-gradient kernels read
every transposing fragment that way until round 5 (100 conversions in the listing, 256 registers; 133 -> 118 us per launch
without them).  A handful per kernel is set-up code; more than GENERIC_LIMIT null checks in one kernel is flagged.

usage: tools/check_flat_ops.py [path/to/libdvt_hip.so]      (exit 1 when a kernel outside ALLOWED has FLAT instructions or
                                                            more than GENERIC_LIMIT generic-pointer null checks)
"""
import os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
GENERIC_LIMIT = 12
ALLOWED_GENERIC = ("patchify_generic",)   # real null checks of optional global pointers in an unrolled loop (any-patch-size fallback)
ALLOWED = ("rocprim",)          # kernel-name substrings that may keep FLAT accesses (library code: the segmented sort of eval_metrics.hip)


def flat_ops(so_path, generic=None):
    """{kernel symbol: count of flat_load / flat_store / flat_atomic instructions} for the gfx950 code objects of so_path;
    generic (a dict, optional) receives {kernel symbol: count of 64-bit null checks (v_cmp_ne_u64 .., 0)}."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        so = shutil.copy(so_path, tmp)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], check=True, capture_output=True, cwd=tmp)
        for name in sorted(os.listdir(tmp)):
            if "gfx950" not in name:
                continue
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(tmp, name)], check=True,
                                 capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1)
                    out.setdefault(cur, 0)
                elif cur is not None and re.match(r"^\s+flat_(load|store|atomic)", line):
                    out[cur] += 1
                elif cur is not None and generic is not None and re.match(r"^\s+v_cmp_ne_u64_e32 vcc, 0,", line):
                    generic[cur] = generic.get(cur, 0) + 1
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "data-efficient-video-transformers_amd", "libdvt_hip.so")
    gen = {}
    ops = flat_ops(so, gen)
    bad = {k: v for k, v in ops.items() if v and not any(a in k for a in ALLOWED)}
    badg = {k: v for k, v in gen.items() if v > GENERIC_LIMIT and not any(a in k for a in ALLOWED + ALLOWED_GENERIC)}
    print(f"{len(ops)} kernels, {sum(1 for v in ops.values() if v)} with FLAT instructions, "
          f"{len(badg)} with more than {GENERIC_LIMIT} generic-pointer null checks")
    for k, v in sorted(bad.items(), key=lambda kv: -kv[1]):
        print(f"  {v:4d} flat  {k}")
    for k, v in sorted(badg.items(), key=lambda kv: -kv[1]):
        print(f"  {v:4d} null checks  {k}")
    return 1 if bad or badg else 0


if __name__ == "__main__":
    sys.exit(main())
