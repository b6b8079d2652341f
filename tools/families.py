import json,sys
for l in open(sys.argv[1]):
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(sys.argv[1], d['value'], d['ms_per_step'])
    r=d.get('roofline')
    if not r: continue
    print('  DOM %-60s %7.3f ms %8.1f %s frac %.3f'%(r['kernel'][:60], r['ms_per_step'], r['achieved'], r['unit'], r['frac']))
    for f in r.get('families',[]):
        print('      %-60s %7.3f ms %8.1f %s frac %.3f'%(f['kernel'][:60], f['ms_per_step'], f['achieved'], f['unit'], f['frac']))
    for f in r.get('families_serial_schedule',[]):
        print('   SER %-60s %7.3f ms %8.1f %s frac %.3f'%(f['kernel'][:60], f['ms_per_step'], f['achieved'], f['unit'], f['frac']))
