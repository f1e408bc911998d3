"""Which NeutrinoBaseNetwork (and therefore which HIP runtime) a holder sub-module belongs to.  Kept outside the modules
(an integer key on the child, weak references here) so that modules stay picklable / deep-copyable."""
import itertools
import weakref

_owners = weakref.WeakValueDictionary()
_keys = itertools.count(1)


def register(child, owner) -> None:
    key = next(_keys)
    _owners[key] = owner
    child._owner_key = key


def owner_of(child):
    return _owners.get(getattr(child, "_owner_key", 0))
