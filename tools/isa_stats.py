"""Per-kernel instruction mix / register use of a hipcc -S listing (development aid)."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for name in re.findall(r"^(_Z[\w]+):", s, re.M):
    if pat not in name:
        continue
    a = s.index("\n" + name + ":")
    b = s.find(".end_amdhsa_kernel", a)
    if b < 0:
        continue
    c = Counter()
    for line in s[a:b].split("\n"):
        line = line.strip()
        if not line or line.startswith((".", ";", "//")) or line.endswith(":"):
            continue
        c[line.split()[0]] += 1
    print(name, sum(c.values()))
    print("  ", c.most_common(12))
    print("  ", re.findall(r"; (?:NumVgprs|ScratchSize|Occupancy|NumAgprs): \d+", s[b:b + 4000]))
