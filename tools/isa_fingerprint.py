#!/usr/bin/env python3
"""Per-kernel fingerprint of a gfx950 assembly dump (hipcc --cuda-device-only -S): a hash of the instruction stream (labels, comments and
debug directives stripped) + the register / spill / LDS numbers of the kernel descriptor.  Used to show that a source refactor of
csrc/gemm.hip left every kernel's machine code unchanged:   python tools/isa_fingerprint.py a.s [b.s]"""
import hashlib, re, sys


def kernels(path):
    out, cur, body = {}, None, []
    meta = {}
    name = None
    for line in open(path, errors="replace"):
        s = line.strip()
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", s)
        if m and not s.startswith(".L"):
            cur, body = m.group(1), []
            continue
        if cur:
            if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
                pass
            if s.startswith(".size") or s.startswith(".Lfunc_end"):
                if body:
                    out[cur] = hashlib.sha256("\n".join(body).encode()).hexdigest()[:16] + f" n={len(body)}"
                cur = None
                continue
            if not s or s.startswith(";") or s.startswith(".") and not s.startswith(".L"):
                continue
            s = re.sub(r"\s*;.*$", "", s)
            s = re.sub(r"\.LBB\d+_\d+", "L", s)
            if s.endswith(":"):
                continue
            body.append(s)
        m = re.match(r"\.name:\s+(\S+)", s)
        if m:
            name = m.group(1); meta.setdefault(name, {})
        m = re.match(r"\.(vgpr_count|vgpr_spill_count|sgpr_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size|agpr_count):\s+(\d+)", s)
        if m and name:
            meta[name][m.group(1)] = int(m.group(2))
    return out, meta


a, ma = kernels(sys.argv[1])
if len(sys.argv) == 2:
    for k in sorted(a):
        print(k, a[k], ma.get(k, {}))
    sys.exit(0)
b, mb = kernels(sys.argv[2])
same = diff = 0
for k in sorted(set(a) | set(b)):
    if a.get(k) == b.get(k) and ma.get(k) == mb.get(k):
        same += 1
    else:
        diff += 1
        print("DIFF", k, a.get(k), b.get(k), ma.get(k), mb.get(k))
print(f"{same} kernels identical (instruction stream + descriptor numbers), {diff} differ")
