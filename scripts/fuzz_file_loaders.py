#!/usr/bin/env python3
"""Mutation fuzz of the file loaders under AddressSanitizer + UBSan (CPU only): the payload of a
.ski built from the fixture genomes and of a fixture .skm is mutated (byte flips, truncation,
insertions), re-framed with correct snappy checksums so that the MessagePack / CBOR / roaring layers see it, and
fed to `sketchlib inverted precluster --count`, `skl_dbtool info` and (the .skm, next to its .skd)
`sketchlib dist --ref-completeness-file`, which indexes per-sample arrays with what the .skm says.  Anything but a clean error
exit (sanitizer report, signal, time-out) is printed.

Build the two sanitised binaries first (from sketchlib.rust_amd/csrc):
  g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined host/*.cpp -I../../include \
      -L_build -lsketchlib_dist_hip -lpthread -lz -o /tmp/sketchlib_asan
  g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined tools/dbtool_main.cpp \
      $(ls host/*.cpp | grep -v -e cli_main -e distances.cpp -e sketch_gpu) -lpthread -lz -o /tmp/dbtool_asan
usage: fuzz_file_loaders.py [seed] [rounds]     (run from the repository root)
Round 1 of this found: an indefinite-length CBOR string read past the end of the input, a string
length that overflowed the bounds check, bitmaps naming samples beyond n_samples (out-of-bounds
writes in the candidate search), run containers leaving their key range, and an hour-long run on
an absurd n_samples -- all refused at load time now (tests/test_fileformat_cpu.py).
"""
import os, random, subprocess, shutil, sys
random.seed(int(sys.argv[1]) if len(sys.argv)>1 else 11)
N=int(sys.argv[2]) if len(sys.argv)>2 else 600
def crc32c_masked(b):
    table=[]
    for i in range(256):
        c=i
        for _ in range(8): c=(c>>1)^0x82F63B78 if c&1 else c>>1
        table.append(c)
    c=0xFFFFFFFF
    for x in b: c=table[(c^x)&0xFF]^(c>>8)
    c^=0xFFFFFFFF
    return ((((c>>15)|(c<<17))&0xFFFFFFFF)+0xA282EAD8)&0xFFFFFFFF
def frame(raw):
    out=b"\xff\x06\x00\x00sNaPpY"
    for o in range(0,max(len(raw),1),60000):   # the format caps a chunk at 65536 uncompressed bytes
        part=raw[o:o+60000]
        body=crc32c_masked(part).to_bytes(4,'little')+part
        out+=b"\x01"+len(body).to_bytes(3,'little')+body
    return out
fx=os.path.abspath('tests/golden/reference_fixtures')
tmp='/tmp/skifuzz'; shutil.rmtree(tmp,ignore_errors=True); os.makedirs(tmp)
env={**os.environ,'ASAN_OPTIONS':'detect_leaks=0','LD_LIBRARY_PATH':'/root/repo/sketchlib.rust_amd/csrc/_build'}
def run(cmd,t=20):
    try:
        r=subprocess.run(cmd,capture_output=True,env=env,cwd=tmp,timeout=t); return r.returncode, r.stderr.decode('utf-8','replace')
    except subprocess.TimeoutExpired:
        return -999,'TIMEOUT'
names=['14412_3#82.contigs_velvet.fa.gz','14412_3#84.contigs_velvet.fa.gz','R6.fa.gz','TIGR4.fa.gz']
for n in names: shutil.copy(os.path.join(fx,n),tmp)
run(['/tmp/sketchlib_asan','inverted','build','-o','idx','-k','21','-s','50',*names],120)
run(['/tmp/dbtool_asan','unframe','idx.ski','rawski'])
raw=open(os.path.join(tmp,'rawski'),'rb').read()
run(['/tmp/dbtool_asan','unframe',os.path.join(fx,'sketches1.skm'),'rawskm'])
rawskm=open(os.path.join(tmp,'rawskm'),'rb').read()
shutil.copy(os.path.join(fx,'sketches1.skd'),os.path.join(tmp,'m.skd'))
open(os.path.join(tmp,'comp.txt'),'w').write(''.join(f"{n}\t0.9\n" for n in names))
def mutate(data):
    b=bytearray(data); kind=random.random()
    if kind<0.7:
        for _ in range(random.randint(1,3)): b[random.randrange(len(b))]=random.randrange(256)
    elif kind<0.85: b=b[:random.randrange(1,len(b))]
    else:
        p=random.randrange(len(b)); b[p:p]=bytes(random.randrange(256) for _ in range(random.randint(1,6)))
    return bytes(b)
seen={}; bad=0
for it in range(N):
    for kind,(payload,cmd) in {'ski':(raw,['/tmp/sketchlib_asan','inverted','precluster','m.ski','--count']),
                               'skm':(rawskm,['/tmp/dbtool_asan','info','m']),
                               'skm+completeness':(rawskm,['/tmp/sketchlib_asan','dist','m','--ref-completeness-file','comp.txt'])}.items():
        open(os.path.join(tmp,'m.'+kind.split('+')[0]),'wb').write(frame(mutate(payload)))
        rc,err=run(cmd)
        if 'Sanitizer' in err or rc<0 or 'runtime error' in err:
            bad+=1
            lines=[l.strip() for l in err.splitlines() if l.strip().startswith('#')][:5]
            key=(kind,rc)+tuple(l.split(' in ')[-1][:90] for l in lines[:3])
            if key not in seen:
                seen[key]=1
                print('---- crash',kind,len(seen),rc); print('\n'.join(lines) if lines else err[-300:])
print('runs',2*N,'bad',bad,'distinct',len(seen))
