"""The reference's Forest (vessel_graph_generation/forest.py): root stumps drawn from the global generators in the constructor, the
grown trees readable through get_trees() / get_nodes() / save() once Greenhouse.develop_forest() has run on the device."""
import csv
import math
import os
import random

import numpy as np

from .arterial_tree import ArterialTree


class Forest:
    def __init__(self, config, d_0, r_0, sim_space, arterial=True, nerve_center=None, nerve_radius=0):
        self.config = config
        self.trees = []
        self.sim_space = sim_space
        self.size_x, self.size_y, self.size_z = self.sim_space.shape
        self.arterial = arterial
        if config["type"] == "nerve":
            self._initialize_tree_stumps_from_nerve(config, d_0, r_0, nerve_center, nerve_radius)
        elif config["type"] == "stumps":
            self._initialize_tree_stumps(config, d_0, r_0)
        else:
            raise NotImplementedError(f"The Forest initialization type '{config['type']}' is not implemented. Try 'stump' or 'nerve' instead.")

    def _name(self, i):
        return f'{"Arterial" if self.arterial else "Venous"}Tree{i}'

    def _initialize_tree_stumps_from_nerve(self, config, d_0, r_0, nerve_center=None, nerve_radius=0):
        """All roots inside the optic-nerve disc: five random.random() draws per tree (reference :38-66)."""
        for i in range(config["N_trees"]):
            alpha = 2 * math.pi * random.random()
            r = nerve_radius * math.sqrt(random.random())
            pos = np.array([r * math.cos(alpha) + nerve_center[1], r * math.sin(alpha) + nerve_center[0],
                            random.random() * self.sim_space.size_z])
            tree = ArterialTree(self._name(i), pos, r_0, self.size_x, self.size_y, self.size_z, self)
            v = np.array([random.random() - 0.5, random.random() - 0.5, 0])
            tree.add_node(position=pos + v / np.linalg.norm(v) * d_0, radius=r_0, parent=tree.root)
            self.trees.append(tree)

    def _initialize_tree_stumps(self, config, d_0, r_0):
        """Roots on the lateral faces: random.choice of the wall, the face position, three numpy uniforms for the direction
        (reference :68-178)."""
        walls = [k for k, v in config["source_walls"].items() if v]
        size = (self.size_x, self.size_y, self.size_z)

        def free(c, axis):      # direction range of a coordinate that must not leave the cuboid
            return np.random.uniform(-1 if c - d_0 > 0 else 0, 1 if c + d_0 < size[axis] else 0)

        for i in range(1, config["N_trees"] + 1):
            wall = random.choice(walls)
            axis, far = "xyz".index(wall[0]), wall[1] == "1"
            a, z = self.sim_space.get_random_valid_position(along_axis=axis, first=not far or axis == 2)   # reference :169: z1 asks for `first` too
            inward = (lambda: np.random.uniform(-1, -0.1)) if far else (lambda: np.random.uniform(0.1, 1))
            if axis == 0:
                position = np.array([self.size_x - 1e-6 if far else 0, a, z])
                direction = np.array([inward(), free(a, 1), free(z, 2)])
            elif axis == 1:
                position = np.array([a, self.size_y - 1e-6 if far else 0, z])
                direction = np.array([free(a, 0), inward(), free(z, 2)])
            else:               # reference :153-181 (a = x, z = y here)
                position = np.array([a, z, self.size_z - 1e-6 if far else 0])
                direction = np.array([free(a, 0), free(z, 1), inward()])
            direction = direction / np.linalg.norm(direction) * d_0
            tree = ArterialTree(self._name(i), tuple(position), r_0, self.size_x, self.size_y, self.size_z, self)
            tree.add_node(position=tuple(position + direction), radius=r_0, parent=tree.root)
            self.trees.append(tree)

    def _stump_array(self):
        """[2 * N_trees][3]: root, stump child per tree -- the node layout the device starts from."""
        return np.array([n.position for t in self.trees for n in t._level_order[:2]], dtype=np.float64)

    def _load_rows(self, rows, kappa):
        """Hand each tree its rows of the exported edge list: a tree starts at the row (stump child -> root)."""
        starts = []
        at = 0
        for t in self.trees:
            child, root = t._level_order[1].position, t.root.position
            hit = np.nonzero((rows[at:, 0:3] == child).all(axis=1) & (rows[at:, 3:6] == root).all(axis=1))[0]
            if len(hit) == 0:
                raise RuntimeError(f"{t.name}: stump edge missing from the simulator's output")
            at += int(hit[0])
            starts.append(at)
            at += 1
        starts.append(len(rows))
        for t, a, b in zip(self.trees, starts[:-1], starts[1:]):
            t._load_rows(rows[a:b], kappa)

    def get_trees(self):
        return self.trees

    def get_nodes(self):
        for tree in self.trees:
            yield from tree.get_tree_iterator(exclude_root=False, only_active=False)

    def get_node_coords(self):
        for node in self.get_nodes():
            yield node.position

    def save(self, save_directory="."):
        name = f'{"Arterial" if self.arterial else "Venous"}Forest'
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, name + ".csv"), "w+") as file:
            writer = csv.writer(file)
            writer.writerow(["node1", "node2", "radius"])
            for tree in self.get_trees():
                for node in tree.get_tree_iterator(exclude_root=True, only_active=False):
                    writer.writerow([node.position, node.get_proximal_node().position, node.radius])
