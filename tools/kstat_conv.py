import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if n.startswith('void conv_kernel') and (', 16, 1,' in n or ', 64, 1,' in n or ', 32, 2,' in n or ', 64, 2,' in n):
        print('%-100s calls %5s avg %8.1f us'%(n[:100], r['Calls'], float(r['AverageNs'])/1e3))
