"""Locations of the configuration files shipped with the build (same relative paths as the reference's)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GENERATOR_CONFIG = os.path.join(ROOT, "docker", "vessel_graph_gen_docker_config.yml")


def load_yaml(path):
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)


def load_generator_config():
    """docker/vessel_graph_gen_docker_config.yml: the 3x3 mm^2 generator config of BASELINE.json configs[0..1]."""
    return load_yaml(GENERATOR_CONFIG)
