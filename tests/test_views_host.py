"""vegs_amd/views.py without a GPU: with one stream the helpers are plain calls in order."""
import torch


def test_view_streams_on_one_stream_is_plain_calls():
    from vegs_amd import views
    vs = views.ViewStreams("cpu", 1)
    out, ev = vs.run(lambda a, b=2: a * b, 4, b=5)
    assert out == 20 and ev is None
    vs.join()
    assert list(views.render_sequence([1, 2, 3], lambda c: {"x": torch.tensor([c])}, "cpu", streams=1))[2]["x"].item() == 3
