"""Read-only views of a simulated tree with the reference's Node / ArterialTree accessors (vessel_graph_generation/arterial_tree.py):
what generate_vessel_graph.py:41-53, Forest.save and the visualisation scripts walk after develop_forest(). The nodes live on the
device while the forest grows; these objects are built from the exported edge rows (level order per tree, the order of
anytree.LevelOrderIter)."""
import numpy as np


class Node:
    def __init__(self, tree, name, position, radius, parent=None, kappa=4):
        self.tree, self.name = tree, name
        self.position = np.array(position)
        self.radius = radius
        self.kappa = kappa
        self.parent = parent
        self.children = []
        if parent is not None:
            parent.children.append(self)
        self.proximal_num_segments = 0 if parent is None else parent.proximal_num_segments + 1

    def __repr__(self):
        return "{} (position: {}, radius: {}, active: {})".format(self.name, self.position, self.radius, self.active)

    @property
    def active(self):
        return self.tree.forest.sim_space.is_valid_position(self.position)

    @property
    def is_root(self):
        return self.parent is None

    @property
    def is_leaf(self):
        return not self.children

    @property
    def is_inter_node(self):
        return self.parent is not None and len(self.children) == 1

    @property
    def is_bifurcation_node(self):
        return len(self.children) == 2

    def _distal(self, child_index):
        if self.is_leaf:
            raise RuntimeError("Unable to analyze distal part. This node does not have any children.")
        if self.is_bifurcation_node:
            if child_index is None:
                raise RuntimeError("Unable to analyze distal part. Unclear which branch to return.")
            return self.children[child_index]
        return self.children[0]

    def get_distal_node(self, child_index=None):
        return self._distal(child_index)

    def get_distal_position(self, child_index=None):
        return self._distal(child_index).position

    def get_distal_radius(self, child_index=None):
        return self._distal(child_index).radius

    def get_distal_segment(self, child_index=None):
        return self._distal(child_index).position - self.position

    def _proximal(self):
        if self.is_root:
            raise RuntimeError("Unable to analyze proximal part. This node is the root.")
        return self.parent

    def get_proximal_node(self):
        return self._proximal()

    def get_proximal_position(self):
        return self._proximal().position

    def get_proximal_radius(self):
        self._proximal()
        return self.radius

    def get_proximal_segment(self):
        return self.position - self._proximal().position


class ArterialTree:
    def __init__(self, name, root_position, r_0, size_x, size_y, size_z, forest):
        self.name = name
        self.init_size_x = self.size_x = size_x
        self.init_size_y = self.size_y = size_y
        self.init_size_z = self.size_z = size_z
        self.r_0 = r_0
        self.scaling_factor = 1.0
        self.forest = forest
        self.root = Node(self, "Root", position=root_position, radius=r_0)
        self.name_counter = 1
        self._level_order = [self.root]

    def add_node(self, position, radius, parent, kappa=4):
        node = Node(self, "Node" + str(self.name_counter), position=position, radius=radius, parent=parent, kappa=kappa)
        self.name_counter += 1
        self._level_order.append(node)
        return node

    def _load_rows(self, rows, kappa):
        """Replace everything below the root by the exported rows of this tree (node xyz, parent xyz, radius; level order)."""
        self.root.children = []
        self._level_order = [self.root]
        self.name_counter = 1
        by_pos = {self.root.position.tobytes(): self.root}
        for row in rows:
            node = self.add_node(row[0:3].copy(), float(row[6]), by_pos[row[3:6].tobytes()], kappa=kappa)
            by_pos[node.position.tobytes()] = node

    def get_tree_iterator(self, exclude_root=False, only_active=False):
        for n in self._level_order:
            if (n.parent is not None or not exclude_root) and (n.active if only_active else True):
                yield n
