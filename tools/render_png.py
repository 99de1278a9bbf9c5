"""Renders a task on the GPU and writes the client's preview of it (client/src/frame.rs:86-103) as a PNG:
python3 tools/render_png.py ROOT_DIR TASK_JSON OUT.png [width height iterations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as P  # noqa: E402

root, task_path, out = sys.argv[1:4]
task = P.RenderTask.from_json(open(task_path).read())
if len(sys.argv) > 5:
    task.camera.resolution = [int(sys.argv[4]), int(sys.argv[5])]
if len(sys.argv) > 6:
    task.config.iterations = int(sys.argv[6])
image = P.Worker(root).render(task)
frame = P.Frame(*task.camera.resolution)
frame.add_render(image)
from PIL import Image  # noqa: E402

Image.fromarray(frame.image(), "RGBA").save(out)
print(out, image.shape, "mean", float(image[image == image].mean()))
