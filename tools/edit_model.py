"""Python model of the K4 block scheme (Myers/Hyyro bit-vector blocks of 64 rows, global distance: the horizontal delta
entering block 0 is +1 in every column; the score is followed at the row of the last pattern base) against the
plain dynamic programme.  Pure Python, small cases; it documents the recurrences the kernel uses."""
import random

M64 = (1 << 64) - 1


def dp(a, b):
    row = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        diag, row[0] = row[0], i
        for j in range(1, len(b) + 1):
            up = row[j]
            row[j] = min(diag + (a[i - 1] != b[j - 1]), up + 1, row[j - 1] + 1)
            diag = up
    return row[len(b)]


def blocks(pat, txt):
    m, n = len(pat), len(txt)
    if m == 0:
        return n
    B = (m + 63) // 64
    peq = [dict() for _ in range(B)]
    for r, c in enumerate(pat):
        peq[r >> 6][c] = peq[r >> 6].get(c, 0) | (1 << (r & 63))
    Pv = [M64] * B
    Mv = [0] * B
    score = m
    lastbit = (m - 1) & 63
    for j in range(n):
        hin = 1
        for b in range(B):
            Eq = peq[b].get(txt[j], 0)
            Xv = Eq | Mv[b]
            if hin < 0:
                Eq |= 1
            Xh = ((((Eq & Pv[b]) + Pv[b]) & M64) ^ Pv[b]) | Eq
            Ph = (Mv[b] | ~(Xh | Pv[b])) & M64
            Mh = Pv[b] & Xh
            if b == B - 1:
                score += ((Ph >> lastbit) & 1) - ((Mh >> lastbit) & 1)
            hout = ((Ph >> 63) & 1) - ((Mh >> 63) & 1)
            Ph = (Ph << 1) & M64
            Mh = (Mh << 1) & M64
            if hin < 0:
                Mh |= 1
            elif hin > 0:
                Ph |= 1
            Pv[b] = (Mh | ~(Xv | Ph)) & M64
            Mv[b] = Ph & Xv
            hin = hout
    return score


if __name__ == '__main__':
    rng = random.Random(5)
    for it in range(400):
        la, lb = rng.randint(0, 200), rng.randint(0, 200)
        alpha = 'ACGT' if it % 3 else 'ACGTN'
        a = ''.join(rng.choice(alpha) for _ in range(la))
        if it % 2:
            b = list(a)
            for _ in range(rng.randint(0, 30)):
                if b and rng.random() < 0.5:
                    b[rng.randrange(len(b))] = rng.choice(alpha)
                elif b and rng.random() < 0.5:
                    del b[rng.randrange(len(b))]
                else:
                    b.insert(rng.randint(0, len(b)), rng.choice(alpha))
            b = ''.join(b)
        else:
            b = ''.join(rng.choice(alpha) for _ in range(lb))
        assert blocks(a, b) == dp(a, b), (a, b)
        assert blocks(b, a) == dp(a, b)
    print('ok')
