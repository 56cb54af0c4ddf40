import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ciri_long_amd import hip
ctx = hip.default_context()
d = tempfile.mkdtemp()
cases = {'empty.fa': b'', 'one.fa': b'>r1\nACGTACGTACGT\n', 'hdr_only.fa': b'>r1', 'nl.fa': b'\n\n', 'one.fq': b'@r1 x\nACGT\n+\nIIII\n', 'crlf.fa': b'>a\r\nACGTAC\r\n>b\r\n\r\n'}
for name, data in cases.items():
    p = os.path.join(d, name)
    open(p, 'wb').write(data)
    for rep in range(2):
        t = ctx.ccs_file(p, name.endswith('.fq'), os.path.join(d, 'o.ccs.fa'), os.path.join(d, 'o.raw.fa'))
    print(name, t, os.path.getsize(os.path.join(d, 'o.ccs.fa')), os.path.getsize(os.path.join(d, 'o.raw.fa')))
ctx.release_file_buffers()
print(ctx.ccs_file(os.path.join(d, 'one.fa'), 0, os.path.join(d, 'o.ccs.fa'), os.path.join(d, 'o.raw.fa')))
print('edge ok')
