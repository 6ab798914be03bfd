"""CLI / HTTP surface contract, following the CODE of the reference where its
tests are stale.  No GPU: the pipelines are mocked exactly like the reference's
test/test_api.py does (MagicMock pipeline, zero latents)."""
from unittest.mock import MagicMock, patch

import pytest
import torch


def test_to_latent_size_rounds_up():
    import txt2image
    assert txt2image.to_latent_size((512, 512)) == (64, 64)
    assert txt2image.to_latent_size((768, 512)) == (96, 64)
    assert txt2image.to_latent_size((513, 513)) == (66, 66)      # code behaviour; the reference's test is stale
    import flux_app
    assert flux_app.to_latent_size((769, 513)) == (98, 66)


def test_cli_flags_and_defaults():
    import txt2image
    a = txt2image.build_parser().parse_args(["a cat"])
    assert (a.model, a.n_images, a.image_size, a.steps, a.guidance, a.n_rows, a.decoding_batch_size, a.output) == (
        "schnell", 4, (512, 512), None, 4.0, 1, 1, "out.png")
    assert a.t5_padding is True and not a.save_raw and not a.quantize and a.seed is None
    a = txt2image.build_parser().parse_args(["x", "--model", "dev", "--image-size", "256x384", "--no-t5-padding", "-q", "-v",
                                             "--save-raw", "--seed", "7", "--n-images", "2", "--steps", "3"])
    assert a.model == "dev" and a.image_size == (256, 384) and a.t5_padding is False and a.quantize and a.verbose
    with pytest.raises(SystemExit):
        txt2image.main(["x", "--steps", "0"])


@pytest.fixture
def client():
    from fastapi.testclient import TestClient
    import flux_app
    return TestClient(flux_app.get_app())


def test_api_static_routes(client):
    r = client.get("/sdapi/v1/sd-models")
    assert r.status_code == 200 and len(r.json()) == 4
    assert set(r.json()[0]) == {"title", "name", "model_name", "hash", "sha256", "filename", "config"}
    assert [m["title"] for m in r.json()] == ["flux-schnell", "flux-dev", "stabilityai/stable-diffusion-2-1-base",
                                              "stabilityai/sdxl-turbo"]
    o = client.get("/sdapi/v1/options").json()
    assert o["sd_backend"] == "Flux MLX" and len(o["sd_model_list"]) == 4
    assert client.post("/sdapi/v1/options", json={"x": 1}).json() == {"success": True}
    p = client.get("/sdapi/v1/progress").json()
    assert p["textinfo"] == "Idle" and p["progress"] == 0 and "state" in p


def test_api_txt2img_with_mock_pipeline(client):
    import flux_app
    pipe = MagicMock()
    pipe.generate_latents.return_value = iter([("cond",), torch.zeros(1, 64, 64), torch.zeros(1, 64, 64)])
    pipe.decode.return_value = torch.zeros(1, 128, 128, 3)
    with patch.object(flux_app.FluxAPI, "init_pipeline", return_value=pipe):
        r = client.post("/sdapi/v1/txt2img", json={"prompt": "test", "width": 128, "height": 128, "steps": 1,
                                                    "cfg_scale": 1.0, "seed": 42, "model": "schnell"})
    assert r.status_code == 200
    body = r.json()
    assert isinstance(body["images"][0], str) and not body["images"][0].startswith("data:")   # raw base64 (code, :199-202)
    assert body["parameters"]["seed"] == 42 and body["info"] == "Generated with Flux schnell model"
    kw = pipe.generate_latents.call_args.kwargs
    assert kw["latent_size"] == (16, 16) and kw["num_steps"] == 1 and kw["seed"] == 42 and kw["n_images"] == 1
    # defaults: steps None -> 2 for schnell; seed -1 -> None
    pipe.generate_latents.return_value = iter([("cond",), torch.zeros(1, 64, 64)])
    with patch.object(flux_app.FluxAPI, "init_pipeline", return_value=pipe):
        client.post("/sdapi/v1/txt2img", json={"prompt": "t"})
    kw = pipe.generate_latents.call_args.kwargs
    assert kw["num_steps"] == 2 and kw["seed"] is None and kw["latent_size"] == (64, 64) and kw["guidance"] == 4.0


def test_api_errors_become_http_500(client):
    import flux_app
    with patch.object(flux_app.FluxAPI, "init_pipeline", side_effect=RuntimeError("boom")):
        r = client.post("/sdapi/v1/txt2img", json={"prompt": "x"})
    assert r.status_code == 500 and r.json()["detail"] == "boom"


def test_sd_routing_uses_sd_signature():
    import flux_app
    pipe = MagicMock()
    pipe.generate_latents.return_value = iter([torch.zeros(2, 8, 8, 4)])
    pipe.decode.side_effect = lambda x, *a: torch.zeros(len(x), 64, 64, 3)     # latents are decoded in batches
    a = flux_app.FluxAPI()
    with patch.object(flux_app.FluxAPI, "init_pipeline", return_value=pipe):
        out = a.generate_images("p", model="stabilityai/sdxl-turbo", batch_size=2, guidance=0.0, return_pil=True)
    kw = pipe.generate_latents.call_args.kwargs
    assert kw["num_steps"] == 2 and kw["n_images"] == 2 and "latent_size" not in kw     # SD ignores width/height (:149-155)
    assert len(out) == 2 and out[0].size == (64, 64) and pipe.decode.call_count == 1


def test_port_helpers():
    import flux_app
    p = flux_app.find_available_port("127.0.0.1", 18760)
    assert flux_app.check_port_available("127.0.0.1", p)
