"""rows of a training curve (scripts/train_synthetic.py --out FILE) identical to profiles/r05_train_curve.json: python scripts/cmp_train_curve.py FILE"""
import json,sys
a=json.load(open('profiles/r05_train_curve.json')); b=json.load(open(sys.argv[1]))
n=min(len(a['log']),len(b['log'])); same=0; first=None
for x,y in zip(a['log'],b['log']):
    if x==y: same+=1
    elif first is None: first=(x['iter'],[k for k in x if x[k]!=y.get(k)])
print("rows identical to the committed curve:",same,"of",n,"first difference:",first)
