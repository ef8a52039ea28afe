"""Instruction histogram of the largest loop of one kernel in a hipcc -S listing: python tools/loop_stats.py file.s substr"""
import re, collections, sys
def loopstats(path, kern_sub, top=45):
    s = open(path).read()
    for f in re.split(r'\n(?=_Z\w+:)', s):
        if f.startswith('_Z') and kern_sub in f.split(':')[0]:
            lines = f.split('\n'); labels = {}
            for n, l in enumerate(lines):
                m = re.match(r'^(\.LBB\d+_\d+):', l)
                if m: labels[m.group(1)] = n
            best = None
            for n, l in enumerate(lines):
                m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
                if m and m.group(1) in labels and labels[m.group(1)] < n:
                    span = (labels[m.group(1)], n)
                    if best is None or span[1] - span[0] > best[1] - best[0]: best = span
            c = collections.Counter()
            for l in lines[best[0]:best[1]]:
                l = l.strip()
                if not l or l.startswith(('.', ';', '/')): continue
                c[l.split()[0]] += 1
            print(f.split(':')[0][:70], 'loop instrs', sum(c.values()))
            print('  ' + '  '.join(f'{k}:{v}' for k, v in c.most_common(top)))
if __name__ == '__main__':
    loopstats(sys.argv[1], sys.argv[2])
