"""Per-kernel register / LDS / wait statistics from a hipcc -S dump (default /tmp/isa/spconv.s)."""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else '/tmp/isa/spconv.s'
pat = sys.argv[2] if len(sys.argv) > 2 else 'sparse_conv'
s = open(path).read()
for m in re.finditer(r'^(_Z\S+): *; @', s, re.M):
    name = m.group(1)
    if pat not in name:
        continue
    b = s[m.start():]
    j = b.find('.end_amdhsa_kernel')
    body, tail = b[:j], b[j:j + 6000]
    cfg = re.findall(r'Li(\d+)E', name)
    g = lambda k: (re.search(r'; ' + k + r': (\d+)', tail) or [None, '?'])[1]
    kind = 'wide' if 'wide' in name else 'base'
    vm0 = len(re.findall(r's_waitcnt vmcnt\(0\)', body))
    print(kind, ','.join(cfg), 'vgpr', g('NumVgprs'), 'agpr', g('NumAgprs'), 'scratch', g('ScratchSize'),
          'lds', g('LDSByteSize'), 'occ', g('Occupancy'), 'mfma', body.count('v_mfma'), 'lines',
          body.count('\n'), 'vmcnt0', vm0)
