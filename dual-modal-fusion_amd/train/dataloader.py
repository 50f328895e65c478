"""train.dataloader — `DataLoaderX`, a DataLoader whose iterator is prefetched by a background thread
(mirror of the reference's train/dataloader.py:6-8, which wraps `prefetch_generator.BackgroundGenerator`;
that package is optional here and a small thread-backed equivalent is used when it is missing)."""
import queue
import threading

from torch.utils.data import DataLoader


class _Background:
    def __init__(self, iterable, max_prefetch=1):
        self.q = queue.Queue(max_prefetch)
        self.t = threading.Thread(target=self._run, args=(iterable,), daemon=True)
        self.t.start()

    def _run(self, iterable):
        try:
            for item in iterable:
                self.q.put((item, None))
            self.q.put((None, StopIteration()))
        except BaseException as e:      # surface loader errors in the consumer
            self.q.put((None, e))

    def __iter__(self):
        return self

    def __next__(self):
        item, err = self.q.get()
        if err is not None:
            raise err
        return item


try:
    from prefetch_generator import BackgroundGenerator
except ImportError:
    BackgroundGenerator = _Background


class DataLoaderX(DataLoader):
    def __iter__(self):
        return BackgroundGenerator(super().__iter__())
