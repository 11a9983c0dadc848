"""Action vocabulary, slip table, map-file lookup and the joint-integer codecs.

Keeps the public names of the reference's ``gym_mapf/envs/__init__.py`` (:3-79).
"""
import os

MAPS_PATH = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', 'maps'))

UP, RIGHT, DOWN, LEFT, STAY = 'UP', 'RIGHT', 'DOWN', 'LEFT', 'STAY'

# index order is part of the joint-action encoding (reference :26)
ACTIONS = [STAY, UP, RIGHT, DOWN, LEFT]
ACTIONS_TO_INT = {name: idx for idx, name in enumerate(ACTIONS)}
ALL_STAY_JOINT_ACTION = 0

# action -> (slip to the agent's right, slip to its left); reference :19-25
POSSIBILITIES = {
    STAY: (STAY, STAY),
    UP: (RIGHT, LEFT),
    RIGHT: (DOWN, UP),
    DOWN: (LEFT, RIGHT),
    LEFT: (UP, DOWN),
}


def map_name_to_files(map_name, scen_id):
    """(<maps>/<name>/<name>.map, <maps>/<name>/<name>-even-<scen_id>.scen); reference :6-10."""
    folder = os.path.join(MAPS_PATH, map_name)
    return (os.path.join(folder, '%s.map' % map_name),
            os.path.join(folder, '%s-even-%s.scen' % (map_name, scen_id)))


def integer_to_vector_multiple_numbers(x, n_options_per_element, n_elements, index_to_element):
    """Mixed-radix decode, least-significant element first; reference :50-67."""
    digits = []
    for i in range(n_elements):
        x, d = divmod(x, n_options_per_element[i])
        digits.append(index_to_element(d))
    return tuple(digits)


def vector_to_integer_multiple_numbers(v, n_options_per_element, element_to_index):
    """Mixed-radix encode: sum_i index(v[i]) * prod_{k<i} radix[k]; reference :70-79."""
    total, weight = 0, 1
    for i, element in enumerate(v):
        total += element_to_index(element) * weight
        weight *= n_options_per_element[i]
    return total


def integer_to_vector(x, options_per_element, n_elements, index_to_element):
    """Reference :32-43 (alias of the multiple-numbers form)."""
    return integer_to_vector_multiple_numbers(x, options_per_element, n_elements, index_to_element)


def vector_to_integer(v, options_per_element, element_to_index):
    """Reference :46-47."""
    return vector_to_integer_multiple_numbers(v, options_per_element, element_to_index)
