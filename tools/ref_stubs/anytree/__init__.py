"""Minimal stand-in for the `anytree` package (absent from the container) so that the reference's
vessel_graph_generation can be IMPORTED, unmodified, by the golden-vector generators in tools/.
Own code, written from anytree's documented behaviour: NodeMixin keeps a parent pointer and an
ordered children list and calls the _pre/_post attach/detach hooks; LevelOrderIter is a BFS that
visits children in insertion order and applies `filter_` to what it yields (not to what it
descends into)."""
from collections import deque


class NodeMixin:
    @property
    def parent(self):
        return getattr(self, "_NodeMixin__parent", None)

    @parent.setter
    def parent(self, value):
        old = self.parent
        if old is value:
            return
        if old is not None:
            self._pre_detach(old)
            old._NodeMixin__children_list().remove(self)
            self._NodeMixin__parent = None
            self._post_detach(old)
        if value is not None:
            self._pre_attach(value)
            value._NodeMixin__children_list().append(self)
            self._NodeMixin__parent = value
            self._post_attach(value)
        else:
            self._NodeMixin__parent = None

    def __children_list(self):
        try:
            return self._NodeMixin__children
        except AttributeError:
            self._NodeMixin__children = []
            return self._NodeMixin__children

    @property
    def children(self):
        return tuple(self._NodeMixin__children_list())

    @property
    def is_leaf(self):
        return len(self._NodeMixin__children_list()) == 0

    @property
    def is_root(self):
        return self.parent is None

    def _pre_attach(self, parent):
        pass

    def _post_attach(self, parent):
        pass

    def _pre_detach(self, parent):
        pass

    def _post_detach(self, parent):
        pass


class LevelOrderIter:
    def __init__(self, node, filter_=None, stop=None, maxlevel=None):
        self.node, self.filter_ = node, filter_

    def __iter__(self):
        q = deque([self.node])
        while q:
            n = q.popleft()
            if self.filter_ is None or self.filter_(n):
                yield n
            q.extend(n.children)


class RenderTree:
    def __init__(self, node):
        self.node = node

    def __str__(self):
        return f"<RenderTree {self.node!r}>"
