"""Minimal stand-in for the third-party `anytree` package (absent from this image, no network).

Harness code only: used in the build container to import the *unmodified* reference from
/root/reference when generating golden vectors (tests/golden/gen_golden.py).  It encodes no
reference logic -- only the public anytree.Node contract the reference relies on:
children are kept in insertion order, `.parent`, `.children` (tuple, settable), `.is_root`, `.name`.
Never imported by the product, never shipped to the GPU box as a dependency of anything.
"""


class Node:
    def __init__(self, name, parent=None, children=None, **kwargs):
        self.name = name
        self._children = []
        self._parent = None
        for k, v in kwargs.items():
            setattr(self, k, v)
        self.parent = parent
        if children:
            self.children = children

    @property
    def parent(self):
        return self._parent

    @parent.setter
    def parent(self, value):
        if self._parent is not None:
            self._parent._children.remove(self)
        self._parent = value
        if value is not None:
            value._children.append(self)

    @property
    def children(self):
        return tuple(self._children)

    @children.setter
    def children(self, value):
        for c in list(self._children):
            c._parent = None
        self._children = []
        for c in value:
            c.parent = self

    @property
    def is_root(self):
        return self._parent is None

    @property
    def is_leaf(self):
        return not self._children


def RenderTree(root):
    def walk(node, depth):
        yield ("  " * depth, "  " * depth, node)
        for c in node.children:
            yield from walk(c, depth + 1)
    return walk(root, 0)
