import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'asurf' not in n: continue
    short=n.split('(')[0].replace('void ','')[:48]
    print('%-50s calls %4s avg %10.1f us  %5s%%' % (short, r['Calls'], float(r['AverageNs'])/1e3, r['Percentage'][:5]))
