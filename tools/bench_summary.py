"""Key numbers of a bench.py JSON line: python tools/bench_summary.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value {:.3e} {} | {:.2f} us/step | roofline frac {:.3f} | kernels {}'.format(d['value'], d['unit'], 1e3 * d['ms_per_step'], d['roofline']['frac'], d.get('kernel_ms')))
if 'sustained' in d: print('sustained {:.3e} | streams {}'.format(d['sustained']['value'], ['{:.3e}'.format(s['value']) for s in d.get('streams', [])]))
others = d.get('other_configs') or []
if isinstance(others, dict): others = [others]
for c in others:
    if 'error' in c: print('  ' + str(c)); continue
    print('  {:<60s} {:.3e} evals/s  frac {:.3f}  oracle err {:.1e}'.format(c['workload'][:60], c['value'], c['roofline']['frac'], c['oracle_check']['max_rel_err_vs_oracle']))
if 'config5_strong' in d: print('config 5: {:.2f} us per update ({:.3e} evals/s), oracle err {:.1e}'.format(d['config5_strong']['us_per_update'], d['config5_strong']['value'], d['config5_strong']['oracle_check']['max_rel_err_vs_oracle']))
if 'chains_weak' in d: print('chains_weak: {:.3e} evals/s'.format(d['chains_weak']['value']))
if d.get('cpu_baseline'): print('cpu_baseline: {:.1f} evals/s on {} core(s)'.format(d['cpu_baseline']['value'], d['cpu_baseline']['cores']))
if d.get('mh_chains') and 'per_config' in d['mh_chains']:
    print('mh_chains: ' + ', '.join('{:d} x {:d}: {:.1f} us per try ({:.2f} M evals/s, acceptance {:.2f})'.format(e['chains'], e['vectorize'], e['us_per_try'], e['value'] / 1e6, e['mean_acceptance_rate'])
                                    for e in d['mh_chains']['per_config']))
