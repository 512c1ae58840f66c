"""Per-layer conv launch durations (one train step of UNet(1,1,bilinear) B=8 512^2) from a rocprofv3 kernel trace."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2])
rows.sort(key=lambda r: int(r['Start_Timestamp']))
conv = [r for r in rows if 'conv3x3' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']]
per = len(conv) // nsteps
L = [('inc.0',1,64,512),('inc.3',64,64,512),('d1.0',64,128,256),('d1.3',128,128,256),('d2.0',128,256,128),('d2.3',256,256,128),('d3.0',256,512,64),('d3.3',512,512,64),('d4.0',512,512,32),('d4.3',512,512,32),('u1.0',1024,512,64),('u1.3',512,256,64),('u2.0',512,256,128),('u2.3',256,128,128),('u3.0',256,128,256),('u3.3',128,64,256),('u4.0',128,64,512),('u4.3',64,64,512)]
seq = [('fwd', l) for l in L]
for l in reversed(L):
    if l[0] != 'inc.0': seq.append(('dgrad', l))
    seq.append(('wgrad', l))
assert per == len(seq), (per, len(seq))
out = {}
for s in range(1, nsteps):
    for (kind, l), r in zip(seq, conv[s * per:(s + 1) * per]):
        out.setdefault((kind, l[0]), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (kind, name), v in out.items():
    if len(sys.argv) > 3 and kind != sys.argv[3]: continue
    print(f"{kind:6s} {name:6s} {sum(v)/len(v):8.1f} us")
red = [r for r in rows if 'bn_relu_bwd_reduce' in r['Kernel_Name']]
print("bn_relu_bwd_reduce per step: n", len(red) / nsteps, "us", sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in red) / nsteps)
