"""Seeds for which the light-sampling draw (raytracer.rs:100) of pixel 0, sample 0, node 0 has the one HIGH word in 2^32 that
leaves `draw > 1 - n_lights * 0.1` open for a single light (threshold 0.9): found offline by
tests/golden/find_light_draw_seeds.c (2^32 Philox calls per seed).  The HIP kernel decides every other draw from the high word
alone — the word attempt 0 of random_in_unit_sphere leaves over — and fetches the low word through a real call only here
(rt_kernel.hip, rt_core.h `light_draw_low_word`); no rendered scene ever meets the case by chance."""
THRESHOLD = 1.0 - 1.0 * 0.1
K = int(THRESHOLD * 2.0 ** 53) + 1          # draw > THRESHOLD  <=>  high * 2^21 + (low >> 11) >= K
OPEN_HIGH_WORD, LOW_PART_BOUND = K >> 21, K & 0x1FFFFF
# seed: the decision the low word makes (True = the hit samples the lights)
OPEN_SEEDS = {5583768346: False, 8527335827: False, 23477444557: True, 29160843763: True, 23630226129: False}


def scene_json(width=2, height=2, spp=1, max_depth=4):
    """One Lambertian sphere that fills the view (so pixel 0's camera ray hits it at node 0) and one light beside the camera."""
    return ('{"width":%d,"height":%d,"samples_per_pixel":%d,"max_depth":%d,"sky":{"texture":""},'
            '"camera":{"look_from":{"x":0.0,"y":0.0,"z":0.0},"look_at":{"x":0.0,"y":0.0,"z":-1.0},"vup":{"x":0.0,"y":1.0,"z":0.0},"vfov":60.0,"aspect":1.0},'
            '"objects":[{"center":{"x":0.0,"y":0.0,"z":-101.0},"radius":100.0,"material":{"Lambertian":{"albedo":[0.8,0.6,0.4]}}},'
            '{"center":{"x":3.0,"y":0.0,"z":-0.5},"radius":0.4,"material":{"Light":{}}}]}' % (width, height, spp, max_depth))
