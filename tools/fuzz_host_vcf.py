"""Randomised comparison of the VCF input path of the host program with the unmodified reference
binary (one comparison individual per run: the reference crashes on the second one): random VCFs
with multi-allelic rows, indels, malformed genotype fields, QUAL values and -q/-v/-M/-w/-D flags;
`ibdgem --plan` against the reference output files.  python tools/fuzz_host_vcf.py [n_cases] [seed]"""
import os, random, subprocess, sys, tempfile
REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
import golden_io as G
from test_host_cli import parse_plan, check_plan_against
REF=os.path.join(REPO,"oracle","_ref","ibdgem"); EXE=os.path.join(REPO,"ibdgem_amd","host","ibdgem")
random.seed(int(sys.argv[2]) if len(sys.argv)>2 else 5)
n_cases=int(sys.argv[1]) if len(sys.argv)>1 else 200
bad=0; compared=0; rows=0
for case in range(n_cases):
    with tempfile.TemporaryDirectory() as d:
        N=random.choice([1,2,5,17]); L=random.randint(1,200)
        names=[f"s{n}" for n in range(N)]
        pos=sorted(random.sample(range(100,100+12*L+50),L))
        chrom=random.choice(["1","chr7"])
        with open(os.path.join(d,"p.vcf"),"w") as fh:
            fh.write("##fileformat=VCFv4.2\n##source=fuzz\n")
            fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t"+"\t".join(names)+"\n")
            for p in pos:
                r=random.random()
                if r<0.05: ref,alt="AT","A"
                elif r<0.10: ref,alt="A","C,G"
                elif r<0.13: ref,alt="a","G"
                else: ref,alt=random.sample("ACGT",2)
                qual=random.choice(["100","29.5","30","0","."])
                gts=[]
                for n in range(N):
                    g=random.random()
                    if g<0.01: gts.append("./.")
                    elif g<0.02: gts.append("0/2")
                    elif g<0.03: gts.append(".")
                    else: gts.append(random.choice("01")+random.choice("/|")+random.choice("01")+random.choice(["",":35",":12:3"]))
                fh.write(f"{chrom}\t{p}\trs{p}\t{ref}\t{alt}\t{qual}\tPASS\t.\tGT\t"+"\t".join(gts)+"\n")
        with open(os.path.join(d,"p.pileup"),"w") as fh:
            for p in pos:
                if random.random()<0.1: continue
                cov=random.choice([0,1,1,2,3,5,9,25])
                bases="".join(random.choices("ACGTacgtN",k=cov)) if cov else "*"
                q="I"*cov if cov else "*"
                fh.write(f"{chrom}\t{p}\tN\t{cov}\t{bases}\t{q}\t{q}\n")
        args=["-V","p.vcf","-P","p.pileup"]
        if random.random()<0.4: args+=["-v"]
        if random.random()<0.4: args+=["-q",random.choice(["30","10","0"])]
        if random.random()<0.4: args+=["-M",random.choice(["3","8","30"])]
        if random.random()<0.5: args+=["-w",random.choice(["2","10","64"])]
        if random.random()<0.3: args+=["-D",random.choice(["0.5","2"])]
        t=random.choice(names); args+=["-s",t]
        out=os.path.join(d,"out"); os.makedirs(out)
        r=subprocess.run([REF,*args,"-O",out],cwd=d,capture_output=True,text=True)
        o=subprocess.run([EXE,*args,"--plan"],cwd=d,capture_output=True,text=True)
        try:
            assert r.returncode==o.returncode,(r.returncode,o.returncode,r.stderr[-300:],o.stderr[-300:])
            if r.returncode==0:
                plan=parse_plan(o.stdout)
                tab=G.TabFile(os.path.join(out,f"UNKWN.{t}.tab.txt")); summ=G.SummaryFile(os.path.join(out,f"UNKWN.{t}.summary.txt"))
                check_plan_against(plan[t],tab,summ); compared+=1; rows+=len(tab.rows)
        except Exception as e:
            bad+=1; print("MISMATCH case",case," ".join(args),repr(e)[:400],flush=True)
            if bad>5: break
print(f"vcf fuzz: {n_cases} cases, {compared} compared ({rows} rows), {bad} failures")
