"""Tabulates hipcc's -Rpass-analysis=kernel-resource-usage remarks: python tools/kernel_resources.py <remarks file>"""
import re
import sys
txt = open(sys.argv[1]).read()
for b in re.split(r'remark: (?:\S+: )?Function Name: ', txt)[1:]:
    name = b.split()[0]
    d = dict(re.findall(r'remark: (?:\S+:\d+:\d+: )?\s*([A-Za-z ]+?)(?: \[[a-z]+/[A-Za-z]+\])?: (\d+)', b))
    short = re.sub(r'^_ZN4rgbl\d+', '', name)[:46]
    print("%-48s VGPR %4s AGPR %3s SGPR %4s occ %2s LDS %6s scratch %s" % (short, d.get('VGPRs'), d.get('AGPRs'), d.get('TotalSGPRs'),
          d.get('Occupancy'), d.get('LDS Size'), d.get('ScratchSize')))
