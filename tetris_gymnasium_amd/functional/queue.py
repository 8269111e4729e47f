"""Queue plug points of the functional surface (reference functional/queue.py).  The HIP engine implements
the stock 7-bag only; these callables are identity tokens that `reset`/`step` check for."""


def create_bag_queue(config, key):  # pragma: no cover - token
    raise NotImplementedError("evaluated on the device; pass it to reset()/step() unchanged")


def bag_queue_get_next_element(config, queue, queue_index, key):  # pragma: no cover - token
    raise NotImplementedError("evaluated on the device; pass it to reset()/step() unchanged")
