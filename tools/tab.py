import re,collections,sys
rows=[l.strip() for l in open(sys.argv[1])]
groups=collections.OrderedDict()
for l in rows:
    m=re.match(r"(\S+)\s+map=(\d) unroll=(\d) ntl=(\d+) nts=(\d+)\s+grid=\s*(\d+) \(([\d.]+)/CU, occ=(\d+)\) :\s+([\d.]+) ms\s+([\d.]+) GB/s",l)
    if not m: print("??",l); continue
    key=(m.group(1),'map'+m.group(2),'u'+m.group(3),'ntl'+m.group(4),'nts'+m.group(5),'occ'+m.group(8))
    groups.setdefault(key,[]).append((int(m.group(6)),float(m.group(10))))
print("GB/s by grid:", [g for g,_ in list(groups.values())[0]])
for k,v in groups.items():
    print(" ".join(k).ljust(40), " ".join(f"{gb:6.0f}" for _,gb in v))
