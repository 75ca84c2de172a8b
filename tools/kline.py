import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l); r=d['roofline']['kernels']
    print('%.0f G  %.3f ms/step | '%(d['value']/1e3,d['ms_per_step'])+'  '.join('%s %.3f'%(k.replace('sg::mfma_stage_',''),v['avg_ms']) for k,v in r.items()))
