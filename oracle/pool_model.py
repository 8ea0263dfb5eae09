"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Pure-Python model of the reference AsyncIOPool's queue discipline (pool.py:64-123) with the
copies reduced to log entries: which slot a request gets, when it is launched, what
`synchronize_pull` returns, how `free_pull` restarts deferred requests and how the push
index wraps.  The shipped pool is checked against this trace.
"""


class PoolModel:
    def __init__(self, pool_size):
        self.pool_size = pool_size
        self.queue = []
        self.pull_index = -1
        self.push_index = -1
        self.log = []

    def async_pull(self, tag):                      # pool.py:64-74
        self.pull_index = (self.pull_index + 1) % self.pool_size
        self.queue.append((self.pull_index, tag))
        if len(self.queue) <= self.pool_size:
            self.log.append(("launch_pull", self.pull_index, tag))

    def synchronize_pull(self):                     # pool.py:83-88
        idx = self.queue[0][0]
        self.log.append(("sync_pull", idx, self.queue[0][1]))
        return idx

    def free_pull(self):                            # pool.py:90-99
        self.queue.pop(0)
        if len(self.queue) >= self.pool_size:
            idx, tag = self.queue[self.pool_size - 1]
            self.log.append(("launch_pull", idx, tag))
        elif len(self.queue) == 0:
            self.pull_index = -1

    def async_push(self, tag):                      # pool.py:101-109
        self.push_index = (self.push_index + 1) % self.pool_size
        self.log.append(("sync_push", self.push_index))
        self.log.append(("launch_push", self.push_index, tag))

    def synchronize_push(self):                     # pool.py:111-123
        for idx in range(self.pool_size):
            self.log.append(("sync_push", idx))
        self.push_index = -1
