"""Progress display in the reference's stderr format (counterpart of CIRI_long/logger.py:10-28):
``[Fri 2026-10-02 23:00:00] [42%  ] [#####....]`` rewritten in place, newline at 100."""
import sys
import time


class ProgressBar(object):
    def __init__(self, width=50):
        self.last_x = -1
        self.width = width

    def update(self, x):
        assert 0 <= x <= 100
        if int(x) == self.last_x:
            return
        self.last_x = int(x)
        filled = int(self.width * (x / 100.0))
        stamp = time.strftime("[%a %Y-%m-%d %H:%M:%S]", time.localtime())
        sys.stderr.write('\r%s [%-5s] [%s]' % (stamp, '%d%%' % int(x), '#' * filled + '.' * (self.width - filled)))
        sys.stderr.flush()
        if x == 100:
            sys.stderr.write('\n')
