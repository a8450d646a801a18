set -e
mkdir -p gpurun_out/tn_deal
for w in pubmed-4p cora-2p citeseer-2p config5-train; do
  python bench.py --workload $w --no-cpu-baseline --no-dealer-streams > gpurun_out/tn_deal/$w.new.json 2> gpurun_out/tn_deal/$w.new.err
  COGNN_NO_DEALER_TN_DEAL=1 python bench.py --workload $w --no-cpu-baseline --no-dealer-streams > gpurun_out/tn_deal/$w.old.json 2> gpurun_out/tn_deal/$w.old.err
done
python - <<'PY'
import json,glob
for w in ['pubmed-4p','cora-2p','citeseer-2p','config5-train']:
    for v in ['new','old']:
        d=json.loads(open('gpurun_out/tn_deal/%s.%s.json'%(w,v)).read().strip().splitlines()[-1])
        print(w,v,'online %.3f offline %.3f incl %.3f'%(d['ms_per_step'],d.get('offline_ms',-1),1e3*d.get('epoch_time_incl_offline_s',-1)))
PY
