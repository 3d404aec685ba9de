"""Prints the lines of one kernel of a hipcc --save-temps .s file that match a regular expression (with their line index inside
the kernel), plus an opcode histogram: python tools/isa_grep.py <file.s> <kernel name substring> [regex]"""
import collections
import re
import sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else r'global_load|vmcnt|global_atomic|ds_write_b128|s_barrier|s_cbranch|^\.?LBB|global_store|s_load_dwordx8|buffer_'
starts = [m for m in re.finditer(r'^(\S+):\s*(?:;.*)?$', s, re.M) if name in m.group(1) and not m.group(1).startswith('.')]
for m in starts:
    end = s.find('s_endpgm', m.end())
    body = s[m.end():end].split('\n')
    print("==", m.group(1), len(body), "lines")
    ops = collections.Counter(l.split()[0] for l in body if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';')))
    print(ops.most_common(30))
    for i, l in enumerate(body):
        if re.search(pat, l.strip()):
            print(i, l.strip()[:120])
