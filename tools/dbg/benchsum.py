import json,sys
d=json.load(open(sys.argv[1])); e=d["extras"]
print(sys.argv[1], "headline", d["value"], "ragged", e["ragged_real_shape"]["registrations_per_s"], e["ragged_real_shape_matched_sizes"]["registrations_per_s"],
      "frame", e["frame_pair"]["max_points_2048"]["ms_per_frame_pair"], e["frame_pair"]["max_points_10000"]["ms_per_frame_pair"],
      "stream4", e["frame_pair"]["max_points_2048"]["stream_ms_per_frame_pair_4_in_flight"], e["frame_pair"]["max_points_10000"]["stream_ms_per_frame_pair_4_in_flight"],
      "four", e["four_batches_in_one_call_registrations_per_s"], "c4", d["config4_on_one_gpu"]["all_8192_pairs"]["registrations_per_s"], d["config4_on_one_gpu"]["shard_of_8_gpus_1024_pairs"]["ms_per_step"])
