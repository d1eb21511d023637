import csv,sys,glob
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'))[-1]
rows=list(csv.DictReader(open(f)))
n=22
tot=lambda keys: sum(float(r['TotalDurationNs']) for r in rows if any(k in r['Name'] for k in keys))/n/1e6
print('conv %.3f ms/step | igemm %.3f | p8 %.3f | small %.3f | wgrad-all %.3f | bn %.3f | total %.3f' % (tot(['conv_igemm','conv3x3']), tot(['conv_igemm']), tot(['conv3x3_w8']), tot(['conv3x3_small']), tot(['wgrad']), tot(['bn_']), sum(float(r['TotalDurationNs']) for r in rows)/n/1e6))
