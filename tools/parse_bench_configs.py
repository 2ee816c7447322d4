import json,sys
s=open(sys.argv[1]).read()
dec=json.JSONDecoder(); i=0
while i < len(s):
    while i < len(s) and s[i].isspace(): i+=1
    if i>=len(s): break
    try:
        o,j=dec.raw_decode(s,i); i=j
        for c in (o if isinstance(o,list) else [o]):
            if isinstance(c,dict) and "config" in c: print(c.get("config"), round(c.get("us_per_pcg_iter"),2))
    except Exception:
        i=s.find('\n',i)+1
        if i==0: break
