"""The reference's SimulationSpace (vessel_graph_generation/simulation_space.py) as the object-API adapters need it: the extent of the
unit cuboid, the voxelised sampling mask and the stump-placement draws of `get_random_valid_position`, taken from the GLOBAL numpy /
CPython generators exactly as the reference takes them. Sink sampling itself (get_candidate_sinks) runs on the device."""
import random
import warnings
from math import ceil, sqrt

import numpy as np

GEOMETRY_SIZE = 76


class SimulationSpace:
    def __init__(self, config, FAZ_center=None, FAZ_radius=None, nerve_center=None, nerve_radius=None):
        self.fixed_geometry = config.get("oxygen_sample_geometry_path") is not None
        if self.fixed_geometry:                                              # reference :29-34
            self.geometry = np.load(config["oxygen_sample_geometry_path"])
            self.geometry_size = max(self.geometry.shape)
            self.shape = np.array(self.geometry.shape) / self.geometry_size
            self.size_x, self.size_y, self.size_z = self.shape
        else:                                                                # reference :36-56
            self.size_x, self.size_y, self.size_z = config["no_voxel_x"], config["no_voxel_y"], config["no_voxel_z"]
            self.shape = np.array([self.size_x, self.size_y, self.size_z])
            assert all(self.shape > 0), "The simulation space dimensions must be postive!"
            if any(self.shape > 1) or all(self.shape != 1):
                warnings.warn("Warning: The largest dimension of the simulation space should be exactly one.")
            self.geometry_size = GEOMETRY_SIZE
            self.FAZ_center = np.array(FAZ_center) * self.geometry_size
            self.FAZ_radius = np.array(FAZ_radius) * self.geometry_size * 0.5
            y, x = np.ogrid[:ceil(self.size_x * self.geometry_size), :ceil(self.size_y * self.geometry_size)]
            self.geometry = (x - self.FAZ_center[0]) ** 2 + (y - self.FAZ_center[1]) ** 2 > self.FAZ_radius ** 2
            if all(nerve_center - nerve_radius <= 1):
                self.nerve_center = np.array(nerve_center) * self.geometry_size
                self.nerve_radius = np.array(nerve_radius) * self.geometry_size
                self.geometry &= (x - self.nerve_center[0]) ** 2 + (y - self.nerve_center[1]) ** 2 > self.nerve_radius ** 2
            else:
                self.nerve_radius = None
                self.nerve_center = None
            self.geometry = np.expand_dims(self.geometry, -1)
        self.valid_voxels = np.argwhere(self.geometry)

    def get_random_valid_position(self, along_axis, first=True):
        """Reference :68-88. With a geometry file: random.choice over the valid voxels of face 0 plus three numpy uniforms; without:
        two numpy uniforms over the wall's extent."""
        if self.fixed_geometry:
            face = np.argwhere(np.take(self.geometry, 0, axis=along_axis))
            index = list(random.choice(face))
            index.insert(along_axis, 0)
            pos = list((np.array(index) + np.random.uniform(0, 1, 3)) / self.geometry_size)
            del pos[along_axis]
            return pos
        if along_axis == 0:
            return np.random.uniform(0, self.size_y), np.random.uniform(0, self.size_z)
        if along_axis == 1:
            return np.random.uniform(0, self.size_x), np.random.uniform(0, self.size_z)
        # reference :82-87: the z branch reads `self.valid_pixels`, which nothing sets -- z walls work with a geometry file only
        raise AttributeError("'SimulationSpace' object has no attribute 'valid_pixels'")

    def is_valid_position(self, pos):
        """Reference :90-99 (the FAZ test compares a unit-cube position with the voxel-scaled centre, as the reference does)."""
        pos = np.asarray(pos)
        if any(pos >= self.shape) or any(pos < 0):
            return False
        if self.fixed_geometry:
            return self.geometry[tuple((pos * self.geometry_size).astype(np.uint16))] > 0
        return sqrt(sum((a - b) ** 2 for a, b in zip(pos, self.FAZ_center))) > self.FAZ_radius
