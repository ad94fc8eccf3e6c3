import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d['config']; ch=c.get('chained',{})
print(sys.argv[1], 'value', d['value'], 'ms', d['ms_per_step'], 'disc', c.get('with_discriminating_ffn',{}).get('volumes_per_s'), 'chained', ch.get('volumes_per_s'), 'seq', ch.get('frame_sequence',{}).get('volumes_per_s'), ch.get('frame_sequence',{}).get('stream_spans_ms'))
