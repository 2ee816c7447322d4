"""Loops of a kernel in a hipcc -S listing: instruction mix per loop body (backward branches).  python tools/isa_loops.py file.s kernel_name_substring"""
import re, sys
def analyze(path, sub):
    lines = open(path).read().split('\n')
    start = None
    for i, ln in enumerate(lines):
        if re.match(r'^_Z\S*' + re.escape(sub) + r'\S*:', ln): start = i; break
    if start is None: print("not found", sub); return
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('\t.section') or lines[i].startswith('.Lfunc_end'))
    order, blocks, cur = ['entry'], {'entry': []}, 'entry'
    for ln in lines[start + 1:end]:
        m = re.match(r'^(\.LBB\d+_\d+):', ln)
        if m: cur = m.group(1); blocks[cur] = []; order.append(cur); continue
        t = ln.strip()
        if ln.startswith('\t') and t and not t.startswith(('.', ';')): blocks[cur].append(t)
    idx = {b: i for i, b in enumerate(order)}
    for b in order:
        for ins in blocks[b]:
            m = re.match(r's_c?branch\w*\s+(\.LBB\d+_\d+)', ins)
            if m and m.group(1) in idx and idx[m.group(1)] <= idx[b]:
                body = []
                for bb in order[idx[m.group(1)]:idx[b] + 1]: body += blocks[bb]
                c = lambda pat: sum(1 for x in body if re.match(pat, x))
                f64 = c(r'v_\w+_f64')
                print("loop %s .. %s: %d insts | valu %d (readlane/writelane %d, dpp %d, f64 %d) salu %d vmem %d lds %d waitcnt %d nop %d" % (m.group(1), b, len(body), c(r'v_'),
                      c(r'v_readlane|v_writelane'), sum(1 for x in body if 'dpp' in x or 'wave_sh' in x), f64, c(r's_(?!waitcnt|nop|cbranch|branch)'), c(r'buffer_|global_'), c(r'ds_'), c(r's_waitcnt'), c(r's_nop')))
analyze(sys.argv[1], sys.argv[2])
