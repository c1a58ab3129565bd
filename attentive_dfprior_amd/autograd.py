"""Autograd bridge for Renderer.render_batch_ray (training path of src/Mapper.py:451-473)."""


def render_with_grad(*args, **kwargs):
    raise NotImplementedError('backward pass not built yet')
