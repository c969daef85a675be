"""Per-kernel summary of a hipcc -save-temps .s file: registers, scratch, MFMA count and every `s_waitcnt vmcnt(N)` in program order
(the counted waits of the ping-pong kernels must be exactly the hand-placed ones: a compiler-inserted vmcnt(0) drains the LDS-DMA queue).
usage: isa_report.py file.s [substring of the kernel name ...]"""
import re, sys
s = open(sys.argv[1]).read()
pats = sys.argv[2:]
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?); -- End function', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pats and not any(p in name for p in pats):
        continue
    tail = s[m.end(): m.end() + 4000]          # the resource comment block follows the function body
    def f(key):
        r = re.search(r'; %s: (\d+)' % key, tail)
        return r.group(1) if r else '?'
    waits = re.findall(r's_waitcnt vmcnt\((\d+)\)', body)
    mixed = len(re.findall(r's_waitcnt vmcnt\(\d+\) ', body))
    print(name)
    print('   vgpr %s agpr %s scratch %s lds %s | mfma %d ds_read %d glds %d buffer_store %d global_store %d global_load %d | vmcnt waits %s' % (
        f('NumVgprs'), f('NumAgprs'), f('ScratchSize'), f('LDSByteSize'), body.count('v_mfma'), len(re.findall(r'\bds_read', body)),
        len(re.findall(r'buffer_load_dwordx4.* lds', body)), len(re.findall(r'buffer_store', body)), len(re.findall(r'global_store', body)),
        len(re.findall(r'global_load', body)), ' '.join(waits)))
