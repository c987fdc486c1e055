import csv, sys, collections, glob
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        tot = collections.defaultdict(lambda: collections.Counter()); cnt = collections.defaultdict(lambda: collections.Counter())
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        for k in tot:
            if "k_gemm" not in k and "k_attn" not in k: continue
            print(d.split("/")[-1], k)
            for c in sorted(tot[k]):
                print(f"   {c:28s} {tot[k][c] / cnt[k][c]:16.0f}")
