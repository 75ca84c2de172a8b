"""Compressed trace of the memory / MFMA / wait instructions of one kernel in a hipcc -S listing.

usage: python tools/isa_ops.py file.s mangled_kernel_name [last_n]
"""
import sys


def main():
    s = open(sys.argv[1]).read()
    name = sys.argv[2]
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    ops = []
    for l in s[i:j].splitlines():
        t = l.strip().split()
        if not t:
            continue
        o = t[0]
        if o.startswith(("global_store", "scratch_", "global_load", "s_waitcnt", "v_mfma", "s_cbranch", "s_barrier", "ds_read", "buffer_")):
            ops.append(o if not o.startswith("s_waitcnt") else o + " " + " ".join(t[1:3]))
    out, prev, cnt = [], None, 0
    for o in ops:
        if o == prev:
            cnt += 1
        else:
            if prev:
                out.append("%s x%d" % (prev, cnt))
            prev, cnt = o, 1
    out.append("%s x%d" % (prev, cnt))
    n = int(sys.argv[3]) if len(sys.argv) > 3 else len(out)
    print("\n".join(out[-n:]))


if __name__ == "__main__":
    main()
