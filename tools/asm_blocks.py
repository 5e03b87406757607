"""Per-basic-block instruction census of one kernel in a hipcc -S listing (MFMA / scratch / global / LDS counts): where do spills and waits sit?
  python tools/asm_blocks.py file.s kernel_name_substring"""
import re, sys
t = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(t) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l))
end = next(i for i in range(start, len(t)) if 's_endpgm' in t[i])
blocks = []; cur = ['<entry>', []]
for l in t[start + 1:end]:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append(cur); cur = [l.strip(), []]
    else:
        cur[1].append(l)
blocks.append(cur)
c = lambda ls, *ks: sum(any(k in l for k in ks) for l in ls)
for name, ls in blocks:
    if c(ls, 'v_mfma') or c(ls, 'scratch_'):
        print('%-12s lines %5d mfma %4d scr_ld %3d scr_st %3d gload %3d lds_dma %3d ds_rd %3d ds_wr %3d accvgpr %3d waitcnt %3d valu %4d' % (
            name, len(ls), c(ls, 'v_mfma'), c(ls, 'scratch_load'), c(ls, 'scratch_store'), c(ls, 'global_load_dword'), c(ls, 'global_load_lds'),
            c(ls, 'ds_read', 'ds_load'), c(ls, 'ds_write', 'ds_store'), c(ls, 'v_accvgpr'), c(ls, 's_waitcnt'),
            sum(1 for l in ls if re.match(r'^\s+v_(?!mfma|accvgpr)', l))))
