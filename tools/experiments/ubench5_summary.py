# summarises tools/ubench5 output: the four fastest (kernel, WS, splits) per N
import re, sys, collections
best = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r'N=\s*(\d+) (\w+)<WS\s*(\d+)> js\s*(\d+) .*?([\d.]+) us/step .*= \s*([\d.]+) %(.*)', l)
    if m:
        best[int(m.group(1))].append((float(m.group(5)), m.group(2), int(m.group(3)), int(m.group(4)), m.group(6)))
        e = re.search(r'max\|a\| = ([\d.e+-]+|nan|-nan)', m.group(7))
        if e and not (float(e.group(1)) < 2e-6): print('!! accuracy', l.strip())
    elif l.strip(): print(l.strip())
for n in sorted(best):
    print(f"N={n:6d}: " + ' | '.join(f"{t:.2f} us {k} WS{w} js{j} {p}%" for t, k, w, j, p in sorted(best[n])[:4]))
