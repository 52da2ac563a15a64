import json,csv,sys,glob
def conv(path, top=18):
    rows=json.load(open(path))
    for r in sorted(rows,key=lambda r:-r['us'])[:top]:
        print(f"{r['call']:20s} {str(r['shape']):46s} {r['us']:7.1f}us {r['gflop']:6.3f}GF {r['gflop']/r['us']*1e3:7.2f}TF/s")
    print("conv total us", round(sum(r['us'] for r in rows),1), "n", len(rows))
def stats(pattern, top=22):
    f=sorted(glob.glob(pattern))[-1]
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print("total kernel ms", tot/1e6)
    for r in rows[:top]:
        print(f"{r['Name'][:80]:80s} n={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:7.1f}us tot={float(r['TotalDurationNs'])/1e6:7.2f}ms {float(r['Percentage']):5.1f}%")
if __name__=="__main__":
    if sys.argv[1]=="conv": conv(sys.argv[2])
    else: stats(sys.argv[2])
