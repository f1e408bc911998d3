#!/bin/bash
# The wide fused 1x1 backward (csrc/bwd1x1_wide.hip) issues its loads by inline assembly, so the compiler does not know their results arrive
# late: a register copy or a spill placed between such a load and its s_waitcnt would save stale data.  This check replays the kernel's main
# loop from the generated ISA -- every vector-memory instruction enters a queue, every s_waitcnt vmcnt(N) retires all but the N youngest --
# and fails if any instruction touches the destination of a load that is still queued, or if a compiler-made vmcnt(0) drains the pipeline.
set -e
cd "$(dirname "$0")/../dune-transformercvn_amd/csrc"
ARCH=${1:-gfx950}; shift || true
FLAGS=${@:--mllvm -amdgpu-mfma-vgpr-form}
hipcc --offload-arch=$ARCH -O3 -std=c++17 -fPIC $FLAGS -S --cuda-device-only bwd1x1_wide.hip -o /tmp/tcvn_wide.s 2>/dev/null
python3 - <<'PY'
import re, sys
text = open('/tmp/tcvn_wide.s').read().split('\n')
def regs(tok):
    out = []
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b', tok):
        if m.group(1): out += [(m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)]
        else: out.append((m.group(4), int(m.group(5))))
    return out
bad = 0
for ns in (2, 3, 4):
    a = next(i for i, l in enumerate(text) if l.startswith('_ZN4tcvn12_GLOBAL__N_118k_bwd1x1_wide_bf16ILi%d' % ns))
    b = next(i for i in range(a, len(text)) if 's_endpgm' in text[i])
    lines = text[a:b]
    hdrs = [i for i, l in enumerate(lines) if 'Loop Header: Depth=1' in l]
    hdr = hdrs[-1]; lab = lines[hdr].split(':')[0]
    end = max(i for i, l in enumerate(lines) if re.search(r's_c?branch\w*\s+' + re.escape(lab) + r'$', l.strip()))
    queue, haz, drains, scratch = [], 0, 0, 0
    for rep in range(2):
        for idx in range(hdr, end):
            l = lines[idx].split(';')[0].strip()
            if not l or l.endswith(':') or l.startswith('.'): continue
            parts = l.split(None, 1); op = parts[0]; args = parts[1] if len(parts) > 1 else ''
            if op == 's_waitcnt':
                m = re.search(r'vmcnt\((\d+)\)', args)
                if m:
                    n = int(m.group(1))
                    if n == 0 and rep == 1: drains += 1
                    if len(queue) > n: queue = queue[len(queue) - n:] if n > 0 else []
                continue
            ops = [x.strip() for x in args.split(',')]
            isvm = op.startswith(('global_load', 'global_store', 'scratch_load', 'scratch_store', 'buffer_', 'global_atomic'))
            if op.startswith('scratch_') and rep == 1: scratch += 1
            used = set(r for o in ops for r in regs(o))
            dst = set(regs(ops[0])) if isvm and op.startswith(('global_load_dword', 'scratch_load')) else set()
            for ql, qd in queue:
                if qd & used and rep == 1:
                    haz += 1
                    if haz < 6: print('  NS=%d line %d: %s touches the result of the load at line %d' % (ns, idx + 1, l[:60], ql + 1))
            if isvm: queue.append((idx, dst))
    print('NS=%d: loop of %d lines, hazards %d, vmcnt(0) drains in the loop %d, scratch operations in the loop %d' % (ns, end - hdr, haz, drains, scratch))
    bad += haz + drains + scratch
sys.exit(1 if bad else 0)
PY
