import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r and "spgemm" in r[0]:
        name = r[0].replace("void grb::", "").split("(")[0]
        print("   %-75s calls %s total %.1f ms" % (name[:75], r[1], int(r[2]) / 1e6))
