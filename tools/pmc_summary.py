"""per-kernel mean of one rocprofv3 --pmc counter (csv output), per dispatch"""
import csv, sys, collections
path = sys.argv[1]
acc = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    k = (r["Kernel_Name"].split("(")[0][:70], r["Grid_Size"], r["Counter_Name"])
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"])
print("kernel,grid,counter,dispatches,mean_value")
for (k, g, c), (n, v) in acc.items():
    print('"%s",%s,%s,%d,%.1f' % (k, g, c, n, v / n))
