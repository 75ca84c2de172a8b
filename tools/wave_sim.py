#!/usr/bin/env python
"""In-order single-wave timing model of one item of an MFMA stage kernel, from a `hipcc -S -DSG_STAMPS` listing.

usage: python tools/wave_sim.py file.s mangled_kernel_name [L_vm=1800] [L_lds=128] [issue=4]

The item loop body (loop header .. back edge) is walked once with every conditional forward branch NOT taken; the
s_memtime stamps of the -DSG_STAMPS build delimit setup / volume / lifts / epilogue.  Model: one instruction per
`issue` cycles; v_mfma_f64_16x16x4 holds the matrix pipe 64 cycles, v_mfma_f64_4x4x4_4b 16; VMEM loads complete in
order L_vm cycles after issue, LDS reads L_lds, scalar loads 200; s_waitcnt blocks until the counters allow.  It
answers one question: with nothing else on the SIMD, where does the wave wait - and how would a different latency
or schedule change that?  (profiles/r03/kernel_experiments.txt compares it with the measured one-wave-per-SIMD
stamps, SEIGEN_HIP_GRID_BLOCKS=256.)"""
import re
import sys


def body(s, name):
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    lines = s[i:j].splitlines()
    # the item loop: from the first label marked "Loop Header" with Depth=1 ... to the last branch back to it
    hdr = None
    for k, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):.*This Loop Header: Depth=1", l)
        if m:
            hdr = (k, m.group(1))
            break
    if hdr is None:
        raise SystemExit("no loop header found")
    # hipcc lays the loop out rotated (latch blocks before the header): walk from the header to the end of the
    # function body; what follows the last stamp is the loop's tail and the exit
    out = []
    for l in lines[hdr[0] + 1:]:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        if t.startswith("s_endpgm"):
            break
        out.append(t)
    return out


def simulate(ins, L_vm=1800, L_lds=128, issue=4, L_sm=200, verbose=False):
    t = 0.0
    pipe_free = 0.0
    vm, lgkm = [], []        # completion times of outstanding ops, in issue order
    phases, stall = [], {"vm": 0.0, "lgkm": 0.0, "mfma": 0.0}
    mark = 0.0
    pstall = dict(stall)
    busy = 0.0
    for raw in ins:
        op = raw.split()[0]
        if op == "s_memtime":
            phases.append((t - mark, {k: stall[k] - pstall[k] for k in stall}, busy))
            mark, pstall, busy = t, dict(stall), 0.0
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", raw)
            if m:
                n = int(m.group(1))
                if len(vm) > n:
                    tt = vm[len(vm) - n - 1]
                    if tt > t:
                        stall["vm"] += tt - t
                        t = tt
                    del vm[:len(vm) - n]
            m = re.search(r"lgkmcnt\((\d+)\)", raw)
            if m:
                n = int(m.group(1))
                if len(lgkm) > n:
                    tt = max(lgkm[:len(lgkm) - n])          # SMEM may return out of order: wait for all of them
                    if tt > t:
                        stall["lgkm"] += tt - t
                        t = tt
                    del lgkm[:len(lgkm) - n]
            t += issue
            continue
        if op.startswith("v_mfma"):
            occ = 16.0 if "4x4x4" in op else 64.0
            if pipe_free > t:
                stall["mfma"] += pipe_free - t
                t = pipe_free
            pipe_free = t + occ
            busy += occ
            t += issue
            continue
        if op.startswith("global_load") or op.startswith("scratch_load") or op.startswith("buffer_load"):
            vm.append(max(t + L_vm, vm[-1] if vm else 0.0))
        elif op.startswith("global_store") or op.startswith("scratch_store") or op.startswith("buffer_store"):
            vm.append(max(t + L_vm, vm[-1] if vm else 0.0))
        elif op.startswith("ds_"):
            lgkm.append(max(t + L_lds, lgkm[-1] if lgkm else 0.0))
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            lgkm.append(t + L_sm)
        if op == "s_nop":
            t += issue * (int(raw.split()[1]) + 1)
        elif op in ("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32"):
            t += 4 * issue
        else:
            t += issue
    phases.append((t - mark, {k: stall[k] - pstall[k] for k in stall}, busy))
    return t, phases


def main():
    s = open(sys.argv[1]).read()
    name = sys.argv[2]
    L_vm = float(sys.argv[3]) if len(sys.argv) > 3 else 1800.0
    L_lds = float(sys.argv[4]) if len(sys.argv) > 4 else 128.0
    issue = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0
    ins = body(s, name)
    n_mfma = sum(1 for x in ins if x.startswith("v_mfma"))
    tot, ph = simulate(ins, L_vm, L_lds, issue)
    print("%d instructions in the item loop, %d MFMAs; L_vm %.0f L_lds %.0f issue %.0f -> %.0f cycles per item" % (len(ins), n_mfma, L_vm, L_lds, issue, tot))
    names = ["(loop top)", "setup", "volume", "lifts", "epilogue", "(tail)"]
    for k, (dt, st, busy) in enumerate(ph):
        print("  %-10s %7.0f cycles   matrix pipe busy %6.0f   waits: vm %6.0f  lgkm %6.0f  pipe %6.0f" % (
            names[k] if k < len(names) else "?", dt, busy, st["vm"], st["lgkm"], st["mfma"]))


if __name__ == "__main__":
    main()
